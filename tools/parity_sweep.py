"""Ad-hoc parity sweep on the GPU box: streaming feature sets on many seeded synthetic objects against
the CPU oracle (test infrastructure; not part of the product path).  Usage: parity_sweep.py [n] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle, parity
from mallorn_astrophysics_amd import synth
from mallorn_astrophysics_amd.columns import COLUMNS
from mallorn_astrophysics_amd.engine import extract_csr
from multiprocessing import get_context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4242
lc = synth.make_lightcurves(n, seed=seed)
TOL = {"stat": dict(rtol=1e-9, atol=1e-12), "tde": dict(rtol=1e-8, atol=1e-8), "color": dict(rtol=1e-9, atol=1e-10),
       "shape": dict(rtol=1e-8, atol=1e-10), "physics": dict(rtol=1e-9, atol=1e-10),
       "research": dict(rtol=1e-8, atol=1e-9)}
INT = {"stat": [c for c in COLUMNS["stat"] if c.endswith("_n_obs") or c == "peak_band"]}

_W = {}


def _init(n_w, seed_w):
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    _W["lc"] = synth.make_lightcurves(n_w, seed=seed_w)


def work(args):
    name, lo, hi = args
    w = _W["lc"]
    return oracle.extract(name, w, w["z"], lo, hi)


def oracle_rows(name, n_w, seed_w):
    """CPU oracle over spawned workers (never forked from a process that holds the GPU runtime)."""
    step = max(1, n_w // 64)
    jobs = [(name, lo, min(lo + step, n_w)) for lo in range(0, n_w, step)]
    saved = {k: os.environ.pop(k) for k in list(os.environ) if k == "LD_PRELOAD" or k.startswith(("ROCP", "HSA_TOOLS"))}
    os.environ["HIP_VISIBLE_DEVICES"] = ""
    for k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):     # read by numpy at import in the workers
        os.environ[k] = "1"
    pool = get_context("spawn").Pool(16, initializer=_init, initargs=(n_w, seed_w))
    os.environ.pop("HIP_VISIBLE_DEVICES", None)
    os.environ.update(saved)
    with pool:
        return np.concatenate(pool.map(work, jobs))

if __name__ == "__main__" and os.path.basename(sys.argv[0]) == "parity_sweep.py":
    for name in ("stat", "tde", "color", "shape", "physics", "research"):
        t0 = time.time()
        got = extract_csr(name, lc, z=lc["z"])
        ref = oracle_rows(name, n, seed)
        tol = TOL[name]
        bad = parity.compare(got, ref, COLUMNS[name], int_cols=INT.get(name, ()), label=name, **tol)
        print(name, n, "objects:", len(bad), "mismatches", f"({time.time() - t0:.1f}s)", flush=True)
        for b in bad[:8]:
            print("   ", b)


def fit_report(name, n_fit):
    """Bounded fits / GP: fraction of (object, column) entries within 1e-4 of the oracle and NaN-mask mismatches
    (no stability probes here -- tests/ hold the probe-based rule; this is the raw agreement at scale)."""
    sub = synth.make_lightcurves(n_fit, seed=seed + 1)
    t0 = time.time()
    got = extract_csr(name, sub, z=sub["z"])
    print(f"{name}: GPU done, running the oracle ...", flush=True)
    ref = oracle_rows(name, n_fit, seed + 1)
    nan_mismatch = int((np.isnan(got) != np.isnan(ref)).sum())
    both = ~np.isnan(got) & ~np.isnan(ref)
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-8)
    close = float((rel[both] <= 1e-4).mean())
    print(f"{name} {n_fit} objects: {100 * close:.2f}% of {int(both.sum())} values within 1e-4, "
          f"{nan_mismatch} NaN-mask mismatches of {got.size} ({time.time() - t0:.0f}s)", flush=True)


if __name__ == "__main__" and os.path.basename(sys.argv[0]) == "parity_sweep.py" and len(sys.argv) > 3:
    for name in sys.argv[3].split(","):
        fit_report(name, int(sys.argv[4]) if len(sys.argv) > 4 else 2000)
