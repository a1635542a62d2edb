#!/usr/bin/env python3
"""Benchmark of the hot path: light curves/s for the feature sets built so far.

One "step" = one pass of the selected feature kernels over one resident batch of synthetic
light curves (SURVEY.md §8d generator).  Inputs are in HBM before the timed region starts.
Multi-GPU: one process per GPU (torch.distributed / RCCL), objects sharded with no data-path
collective, one gather of the feature rows to rank 0 per step; weak scaling (fixed objects/GPU).

Prints ONE JSON line on rank 0 (contract in the task description).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The engine runs its feature sets on four streams; HIP maps streams onto 4 hardware queues by default, and a process
# that also holds RCCL's streams then has two of the engine's streams sharing a queue -- the GP and the fit kernels ran
# one after the other (1.98 s per pass instead of 1.81 s).  Must be in the environment before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
FP64_MFMA_PEAK_TFLOPS = 78.6   # dense fp64 matrix peak of MI355X: v_mfma_f64_16x16x4 = 2048 flop / 64 cycles / SIMD x 1024 SIMDs x 2.4 GHz
                               # (= the fp64 vector rate; the guide's table has no fp64 row)
NCOLS = {"stat": 123, "bazin": 52, "powerlaw": 27, "tde": 25, "color": 83, "shape": 65, "physics": 32, "gp2d": 27,
         "gp1d": 21, "research": 40}


_CPU_LC = None      # sample batch of a CPU-baseline worker


def _cpu_init(path):
    global _CPU_LC
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    g = np.load(path)
    _CPU_LC = {k: g[k] for k in g.files}


def _cpu_worker(args):
    name, lo, hi = args
    import oracle
    return oracle.extract(name, _CPU_LC, _CPU_LC["z"], lo, hi).shape[0]


def _take_objects(lc, n):
    off = lc["offsets"][:n + 1]
    out = {"offsets": np.ascontiguousarray(off), "z": np.ascontiguousarray(lc["z"][:n])}
    for k in ("t", "flux", "err", "band"):
        out[k] = np.ascontiguousarray(lc[k][:off[-1]])
    return out


def usable_cores():
    """Host cores this process may actually use: the scheduler affinity mask and the cgroup CPU quota of the box
    (a GPU box hands a one-GPU job a share of the host, e.g. 16 of its cores), not the host's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, min(n, 64))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sets, lc, budget_s=12.0):
    """Time the oracle (kind "port") on the host cores over a bounded sample of the same workload."""
    import multiprocessing as mp
    import oracle

    cores = usable_cores()
    n_obj = len(lc["offsets"]) - 1
    pilot = min(6, n_obj)
    small = _take_objects(lc, pilot)
    t0 = time.perf_counter()
    for s in sets:
        oracle.extract(s, small, small["z"], 0, pilot)
    per_obj = (time.perf_counter() - t0) / pilot
    sample = int(max(cores * 2, min(n_obj, budget_s * cores / max(per_obj, 1e-6))))
    sample = min(sample, n_obj)
    # fresh (spawned) worker processes that never see the GPU runtime or the parent's BLAS threads;
    # the sample travels through a file, not through per-job pickles
    import tempfile
    tmp = tempfile.NamedTemporaryFile(suffix=".npz", delete=False)
    tmp.close()
    np.savez(tmp.name, **_take_objects(lc, sample))
    chunk = max(1, sample // (cores * 4))
    jobs = [(s, lo, min(lo + chunk, sample)) for s in sets for lo in range(0, sample, chunk)]
    # the workers are plain CPU processes: keep profiler / GPU-tool preloads out of their environment
    saved = {k: os.environ.pop(k) for k in list(os.environ)
             if k == "LD_PRELOAD" or k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS", "ROCTX"))}
    os.environ["HIP_VISIBLE_DEVICES"] = ""
    # one BLAS/OpenMP thread per worker process: the variables must be in the environment the workers
    # START with (numpy reads them at import), else `cores` processes x `cores` threads oversubscribe the host
    thread_vars = ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS")
    saved_threads = {k: os.environ.get(k) for k in thread_vars}
    for k in thread_vars:
        os.environ[k] = "1"
    pool_cm = mp.get_context("spawn").Pool(cores, initializer=_cpu_init, initargs=(tmp.name,))
    os.environ.pop("HIP_VISIBLE_DEVICES", None)
    os.environ.update(saved)
    with pool_cm as pool:
        pool.map(_cpu_worker, jobs[:cores])            # untimed: process start-up + imports
        t0 = time.perf_counter()
        pool.map(_cpu_worker, jobs)
        dt = time.perf_counter() - t0
    for k, v in saved_threads.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    os.unlink(tmp.name)
    return {"value": sample / dt, "unit": "light curves/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
            "sample": f"first {sample} objects of the benchmark batch, sets {'+'.join(sets)}, "
                      f"multiprocessing.Pool({cores}) over the numpy/scipy oracle",
            "single_core_est": 1.0 / per_obj}


def local_shard(objects, seed, rank, world, sets):
    """This rank's shard of ONE synthetic survey of world x `objects` light curves: block b of the survey is
    make_lightcurves(objects, seed + b) (objects are independent, so the survey is the concatenation of its blocks);
    the survey is cut into `world` contiguous shards by dist.shard_bounds -- the cost-aware (a N + c N^2 + b N^3)
    sharding users get from dist.extract_sharded.  A rank generates only the blocks its shard touches.  Per-GPU work
    stays fixed as world grows: weak scaling.  Returns (csr, bounds, n_local)."""
    from mallorn_astrophysics_amd import synth
    from mallorn_astrophysics_amd.dist import shard_bounds
    n_all = np.concatenate([synth.lengths(objects, seed=seed + b) for b in range(world)])
    offsets_all = np.concatenate([[0], np.cumsum(n_all)]).astype(np.int64)
    bounds = shard_bounds(offsets_all, world, sets)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    parts = []
    for b in range(lo // objects, (max(hi, lo + 1) - 1) // objects + 1):
        blk = synth.make_lightcurves(objects, seed=seed + b)
        parts.append(synth.slice_objects(blk, max(lo - b * objects, 0), min(hi - b * objects, objects)))
    lc = parts[0] if len(parts) == 1 else synth.concat(parts)
    n_local = hi - lo
    assert len(lc["offsets"]) - 1 == n_local and np.array_equal(np.diff(lc["offsets"]), n_all[lo:hi])
    return lc, bounds, n_local


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--objects", type=int, default=125000, help="objects per GPU (config 5 = 1M over 8 GPUs)")
    ap.add_argument("--sets", default="", help="comma list; default = every set the library implements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=1000000)
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU with torch.distributed.run as a CHILD
        # process (this process has not touched the GPU yet, and never will) and return its exit code
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print("[bench] launching: " + " ".join(cmd), file=sys.stderr, flush=True)
        raise SystemExit(subprocess.run(cmd).returncode)

    # stdout carries the ONE JSON line and nothing else: whatever a library prints there (RCCL announces its version on
    # stdout when the first communicator is made) goes to stderr instead
    json_fd = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from mallorn_astrophysics_amd import _lib, synth
    from mallorn_astrophysics_amd.columns import SET_NAMES
    from mallorn_astrophysics_amd.engine import DeviceBatch, mask_of

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: a {world}-rank run must not be reported as {a.gpus} GPUs")
    torch.cuda.set_device(local)
    # LCFE_BENCH_FORCE_DIST=1: one-rank rehearsal of the RCCL path on a single-GPU box
    use_dist = world > 1 or os.environ.get("LCFE_BENCH_FORCE_DIST") == "1"
    if use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    lib = _lib.load()
    impl = lib.lcfe_implemented_mask()
    sets = [s for s in a.sets.split(",") if s] or [SET_NAMES[i] for i in range(8) if impl >> i & 1]   # the v34a / v55 workload: sets 0..7 (the per-band sklearn GP, set 8, is opt-in)
    mask = mask_of(sets)
    ncol = int(lib.lcfe_ncols(mask))

    lc, bounds, n_local = local_shard(a.objects, a.seed, rank, world, sets)
    batch = DeviceBatch(lc, z=lc["z"], device=local)
    # two output buffers: the RCCL gather of step k (its own stream) overlaps the kernels of step k+1
    # the gather moves equal blocks (RCCL has no ragged gather): every rank pads its rows to the largest shard
    pad = int(np.diff(bounds).max())
    outs = [torch.full((pad, ncol), float("nan"), dtype=torch.float64, device=batch.device) for _ in range(2)]
    out = outs[0]
    gathered = [torch.empty_like(out) for _ in range(world)] if (use_dist and rank == 0) else None
    pending = [None, None]
    counter = [0]

    nst = int(lib.lcfe_nstatus(mask))
    status = torch.zeros((n_local, nst), dtype=torch.int32, device=batch.device) if nst else None

    def step(prof=False):
        b = counter[0] & 1
        counter[0] += 1
        if pending[b] is not None:
            pending[b].wait()                 # the buffer's previous gather has to be done before it is rewritten
            pending[b] = None
        r = batch.run(mask, out=outs[b][:n_local], status=status, prof=prof)
        if use_dist:
            pending[b] = dist.gather(outs[b], gathered, dst=0, async_op=True)
        return r[2] if prof else None

    def fence():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    note(f"{n_local} objects resident on rank 0 ({a.objects * world} in the survey, {world} shard(s)), sets {'+'.join(sets)}")
    for _ in range(a.warmup):
        step()
    fence()
    note("warmup done")
    kernel_ms = np.zeros(len(SET_NAMES))
    t0 = time.perf_counter()
    busy = 0.0                                # this rank's own kernel time: a step with prof drains the rank's stream
    for _ in range(a.steps):
        ts = time.perf_counter()
        p = step(prof=True)
        busy += time.perf_counter() - ts
        kernel_ms += np.array(p["kernel_ms"])
        note(f"step {_ + 1}/{a.steps}")
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=batch.device)
    # per-rank diagnostics of a multi-GPU run: own step time (without waiting for the others) and shard size
    from mallorn_astrophysics_amd.dist import object_costs
    mine = torch.tensor([busy / a.steps * 1e3, float(n_local), float(lc["offsets"][-1]),
                         float(object_costs(lc["offsets"], sets).sum()) * 1e3], dtype=torch.float64, device=batch.device)
    per_rank = [mine]
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        per_rank = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
    dt = float(tmax.item())
    if rank != 0:
        dist.destroy_process_group()
        return
    per_rank = np.array([x.cpu().numpy() for x in per_rank])

    n_pts = int(lc["offsets"][-1])
    kernel_ms /= a.steps
    per_set = {s: float(kernel_ms[SET_NAMES.index(s)]) for s in sets}
    # roofline of the statistics kernel if it ran (the kernel BASELINE.json names for the HBM
    # roofline), else of the slowest kernel: algorithmic bytes = 25 B/point + 8 B/object offset
    # + 8 B x F output columns, divided by the HIP-event time of that kernel.
    rk = "stat" if "stat" in sets else max(per_set, key=per_set.get)
    alg_bytes = 25 * n_pts + 8 * (n_local + 1) + 8 * n_local * NCOLS[rk] + (8 * n_local if rk == "physics" else 0)
    achieved = alg_bytes / (per_set[rk] * 1e-3) / 1e9 if per_set[rk] > 0 else 0.0
    # HBM traffic of the roofline kernel from PMC counters (collected offline in separate rocprofv3
    # --pmc passes on this exact workload; bench.py cannot run under the counters itself)
    traffic = None
    traffic_src = None
    for rr in ("r03", "r02"):
        tj = os.path.join(ROOT, "profiles", f"{rr}_stat_traffic.json")
        if rk == "stat" and world == 1 and a.objects == 125000 and a.seed == 1000000 and os.path.exists(tj):
            traffic = json.load(open(tj)).get("hbm_bytes_per_pass")
            traffic_src = f"profiles/{rr}_stat_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on this workload)"
            break
    # the kernel that decides `value`: the 2-D GP.  Flops from the status words of the last pass: every L-BFGS-B
    # evaluation sweeps the N x N Gram matrix (N^3 flops on the fp64 MFMA: lower-triangle tiles, N/16 pivot steps) and
    # builds the Gram matrix and the gradient (30 N^2); the prediction pass at the optimum reuses the last evaluation's
    # alpha (an extra sweep only when L-BFGS-B stepped back to an earlier iterate: not counted).
    extra = {}
    if "gp2d" in sets and status is not None:
        st0 = 0
        for sname in sets:
            if sname == "gp2d":
                break
            st0 += int(lib.lcfe_nstatus(mask_of([sname])))
        stg = status[:, st0:st0 + 4].cpu().numpy().astype(np.float64)
        n_eval, n_valid = stg[:, 2], stg[:, 3]
        fitted = n_eval > 0
        flops = float((n_eval * n_valid ** 3 + n_eval * 30.0 * n_valid ** 2)[fitted].sum())
        gp_ms = per_set["gp2d"]
        extra["gp2d"] = {"kernel": "gp_kernel<NP> tiers (L-BFGS-B over a blocked symmetric sweep on v_mfma_f64_16x16x4)",
                         "bound": "mfma", "flops": flops, "evaluations": float(n_eval.sum()), "objects_fitted": int(fitted.sum()),
                         "ms": gp_ms, "achieved": flops / (gp_ms * 1e-3) / 1e12 if gp_ms > 0 else 0.0, "peak": FP64_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": flops / (gp_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS if gp_ms > 0 else 0.0,
                         "note": "flops = sum over objects of n_eval (N^3 + 30 N^2) (N = valid points, n_eval from the "
                                 "status words); ms = the set's HIP-event time (it shares the chip with the fit kernels unless LCFE_SERIAL=1)"}
    res = {
        "metric": "light curves/sec", "value": a.objects * world * a.steps / dt, "unit": "light curves/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"config-5 shard: {'+'.join(sets)} feature sets ({ncol} columns) on "
                               f"{a.objects} synthetic 6-band light curves per GPU ({n_local} objects, {n_pts} points on rank 0)",
                   "objects_per_gpu": a.objects, "survey_objects": a.objects * world, "sets": sets,
                   "parallelism": (f"one {a.objects * world}-object survey cut into {world} cost-balanced contiguous shards "
                                   "(dist.shard_bounds), one RCCL gather of the feature rows per step") if world > 1 else "single GPU"},
        "kernel_ms": per_set,
        "kernel_ms_note": ("per-set HIP-event times on one stream (LCFE_SERIAL=1)" if os.environ.get("LCFE_SERIAL") == "1" else
                           "statistics (+ binning) runs alone; the other sets run concurrently on forked streams, "
                           "so their event times overlap and do not add up to ms_per_step"),
        "roofline": {"kernel": "statistics set: bin_kernel, stat_plan_kernel, stat_lanes_all_kernel, stat_lean_kernel<256|512>, fallback launch" if rk == "stat" else f"set_kernel<{rk}>", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg_bytes, "launch_ms": per_set[rk]},
        "roofline_extra": extra,
        "ranks": {"step_ms_min_mean_max": [float(per_rank[:, 0].min()), float(per_rank[:, 0].mean()), float(per_rank[:, 0].max())],
                  "per_rank": [{"rank": r, "step_ms": float(per_rank[r, 0]), "objects": int(per_rank[r, 1]), "points": int(per_rank[r, 2]),
                                "predicted_ms": float(per_rank[r, 3])} for r in range(len(per_rank))],
                  "note": "step_ms = a rank's own kernel time per step (host clock around the engine call, which drains the "
                          "rank's stream); predicted_ms = dist.object_costs summed over the rank's shard"},
    }
    if not a.no_cpu_baseline:
        note("timing the CPU oracle on the host cores")
        res["cpu_baseline"] = cpu_baseline(sets, lc)
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(res) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
