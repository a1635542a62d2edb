"""Multi-GPU sharding of the hot path.

Objects are independent (the reference's extractors are plain loops over objects), so the batch is
cut into ``world`` contiguous CSR slices balanced by predicted cost (a N + c N^2 + b N^3 per object when the GP is
on, a N otherwise), every rank runs the kernels on
its own GPU with no data-path collective, and ONE gather of the ``[n_local, F]`` float64 blocks to
rank 0 ends the run (``torch.distributed``: backend ``nccl`` = RCCL over xGMI on the GPU node,
``gloo`` in the CPU tests).  Contiguous shards keep the original object order, so no permutation
vector is needed to reassemble the frame.
"""
from __future__ import annotations

import os

import numpy as np


# Cost model of one object with N rows, in GPU-seconds of one MI355X (a N + b N^3 when the GP is on;
# an N^2 term covers the latency-bound part of the small GP tiers).  Coefficients fitted to the per-tier kernel
# times of profiles/r02_bench_serial_kernel_stats.csv (125,000 objects): streaming sets + bounded fits 0.81 s per
# 16.95 M points; GP tiers 64/112/160/240/480+768 rows: 0.021/0.148/0.315/0.391/0.636 s for 13/42/34/25/10 k objects.
COST_POINT = 4.8e-8
COST_GP_N2 = 4.6e-10
COST_GP_N3 = 5.1e-13
GP_SETS = ("gp2d", "gp1d")


def object_costs(offsets, sets=None):
    """Predicted cost of every object for the feature sets `sets` (None = the full v34a/v55 workload)."""
    n = np.diff(np.asarray(offsets, np.int64)).astype(np.float64)
    cost = COST_POINT * n
    if isinstance(sets, (int, np.integer)):                 # a feature-set mask, as mask_of / DeviceBatch.run accept
        from .engine import sets_of
        sets = sets_of(int(sets))
    if sets is None or any(s in GP_SETS for s in ([sets] if isinstance(sets, str) else sets)):
        cost = cost + COST_GP_N2 * n * n + COST_GP_N3 * n * n * n
    return cost


def shard_bounds(offsets, world: int, sets=None, cost=None):
    """Object index bounds [b_0=0, b_1, ..., b_world=n_obj] of ``world`` contiguous shards with (nearly) equal
    predicted COST: a N per object for the streaming sets and the bounded fits, plus c N^2 + b N^3 when a GP set
    is among ``sets`` -- the heavy N^3 tail of the Gram-matrix factorisations decides the slowest rank, not the
    point count.  ``cost`` overrides the model with explicit per-object costs."""
    offsets = np.asarray(offsets, np.int64)
    n_obj = len(offsets) - 1
    if n_obj <= 0:
        return np.zeros(world + 1, np.int64)
    c = object_costs(offsets, sets) if cost is None else np.asarray(cost, np.float64)
    cum = np.concatenate([[0.0], np.cumsum(c)])
    total = cum[-1]
    if total <= 0:                                          # all-empty objects: split by count
        cuts = (np.arange(1, world) * n_obj) // world
    else:
        targets = np.arange(1, world) * (total / world)
        # the boundary whose cumulative cost is nearest to the target
        hi = np.searchsorted(cum, targets, side="left")
        hi = np.clip(hi, 1, n_obj)
        lo = hi - 1
        cuts = np.where(np.abs(cum[lo] - targets) <= np.abs(cum[hi] - targets), lo, hi)
    b = np.concatenate([[0], np.minimum(cuts, n_obj), [n_obj]]).astype(np.int64)
    return np.maximum.accumulate(b)


def shard_csr(csr, rank: int, world: int, z=None, sets=None):
    """The CSR slice (and redshifts) of ``rank``; ``(sub_csr, sub_z, (lo, hi))``."""
    b = shard_bounds(csr["offsets"], world, sets)
    lo, hi = int(b[rank]), int(b[rank + 1])
    off = np.asarray(csr["offsets"], np.int64)
    s, e = int(off[lo]), int(off[hi])
    sub = {"offsets": np.ascontiguousarray(off[lo:hi + 1] - s)}
    for k in ("t", "flux", "err", "band"):
        sub[k] = np.ascontiguousarray(csr[k][s:e])
    return sub, (None if z is None else np.ascontiguousarray(np.asarray(z)[lo:hi])), (lo, hi)


def gather_rows(local, n_total: int, bounds, group=None, dst: int = 0):
    """Gather the per-rank row blocks (torch tensors ``[n_local, F]``, same device type on every
    rank) to ``dst``; returns the ``[n_total, F]`` tensor on ``dst`` and ``None`` elsewhere.

    One collective: ``dist.gather`` on blocks padded to the largest shard (RCCL has no ragged
    gather; the padding rows are dropped on arrival)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [int(bounds[r + 1] - bounds[r]) for r in range(world)]
    pad = max(sizes)
    ncol = local.shape[1]
    buf = local
    if local.shape[0] != pad:
        fill = float("nan") if local.dtype.is_floating_point else 0
        buf = torch.full((pad, ncol), fill, dtype=local.dtype, device=local.device)
        buf[:local.shape[0]] = local
    gl = [torch.empty((pad, ncol), dtype=local.dtype, device=local.device) for _ in range(world)] if rank == dst else None
    dist.gather(buf.contiguous(), gl, dst=dst, group=group)
    if rank != dst:
        return None
    out = torch.empty((n_total, ncol), dtype=local.dtype, device=local.device)
    for r in range(world):
        out[int(bounds[r]):int(bounds[r + 1])] = gl[r][:sizes[r]]
    return out


def extract_sharded(sets, csr, z=None, group=None, return_status=False):
    """Run feature sets on this rank's shard (on torch's current device) and gather to rank 0.

    Every rank passes the SAME full ``csr`` (or at least the same ``offsets``); returns the
    ``[n_obj, ncols]`` numpy matrix on rank 0 and ``None`` on the other ranks (with ``return_status`` the pair
    ``(matrix, status)``; ``status`` is ``None`` for masks without status words).  The collective runs on device
    tensors with the ``nccl`` backend (RCCL over xGMI) and on host copies with ``gloo``."""
    import torch
    import torch.distributed as dist

    from .engine import DeviceBatch

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_obj = len(csr["offsets"]) - 1
    bounds = shard_bounds(csr["offsets"], world, sets)
    sub, sub_z, _ = shard_csr(csr, rank, world, z, sets)
    batch = DeviceBatch(sub, z=sub_z, device=torch.cuda.current_device())
    out, status = batch.run(sets)
    host = dist.get_backend(group) == "gloo"
    full = gather_rows(out.cpu() if host else out, n_obj, bounds, group=group)
    full = None if full is None else full.cpu().numpy()
    if not return_status:
        return full
    full_st = None
    if status is not None:
        full_st = gather_rows(status.cpu() if host else status, n_obj, bounds, group=group)
        full_st = None if full_st is None else full_st.cpu().numpy()
    return (full, full_st) if rank == 0 else None


def extract_multi_gpu(sets, csr, z=None, ngpu=2, backend=None, timeout_s=None):
    """``extract_sharded`` for a caller that is ONE ordinary process (the entry-point scripts, ``extract_all(ngpu=N)``):
    starts ``ngpu`` child ranks with ``torch.distributed.run`` -- new processes, so none of them has touched a GPU
    before it selects its own -- hands them the batch through memory-mapped ``.npy`` files, and returns rank 0's
    ``(matrix, status)``.  Backend ``nccl`` (RCCL) when the node has a GPU per rank, else ``gloo`` with the ranks
    sharing the visible GPUs round-robin (``LCFE_DIST_BACKEND`` overrides)."""
    import shutil
    import socket
    import subprocess
    import sys
    import tempfile

    from . import _lib
    from .engine import mask_of
    from .packing import check_csr

    mask = mask_of(sets)
    n_obj, _ = check_csr(csr)
    lib = _lib.load()
    ndev = lib.lcfe_device_count()
    if ndev < 1:
        raise _lib.LcfeError("extract_multi_gpu: no HIP device visible (lcfe has no CPU fallback)")
    if backend is None:
        backend = os.environ.get("LCFE_DIST_BACKEND") or ("nccl" if ndev >= ngpu else "gloo")
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    tmp = tempfile.mkdtemp(prefix="lcfe_dist_", dir=shm)
    try:
        for k in ("offsets", "t", "flux", "err", "band"):
            np.save(os.path.join(tmp, f"{k}.npy"), np.ascontiguousarray(csr[k]))
        if z is not None:
            np.save(os.path.join(tmp, "z.npy"), np.ascontiguousarray(z, np.float64))
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker.py")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(ngpu)}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), worker, tmp, str(mask), backend]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL across processes on this driver)
        env.setdefault("GPU_MAX_HW_QUEUES", "8")
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
        if r.returncode != 0 or not os.path.exists(os.path.join(tmp, "out.npy")):
            raise _lib.LcfeError(f"extract_multi_gpu: the {ngpu}-rank run failed (rc {r.returncode}):\n"
                                 + (r.stdout + r.stderr)[-4000:])
        out = np.load(os.path.join(tmp, "out.npy"))
        st_path = os.path.join(tmp, "status.npy")
        status = np.load(st_path) if os.path.exists(st_path) else None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if out.shape[0] != n_obj:
        raise _lib.LcfeError(f"extract_multi_gpu: {out.shape[0]} rows came back for {n_obj} objects")
    return out, status
