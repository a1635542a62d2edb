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
       "shape": dict(rtol=1e-8, atol=1e-10), "physics": dict(rtol=1e-9, atol=1e-10)}
INT = {"stat": [c for c in COLUMNS["stat"] if c.endswith("_n_obs") or c == "peak_band"]}

def work(args):
    name, lo, hi = args
    return oracle.extract(name, lc, lc["z"], lo, hi)

if __name__ == "__main__":
    for name in ("stat", "tde", "color", "shape", "physics"):
        t0 = time.time()
        got = extract_csr(name, lc, z=lc["z"])
        step = max(1, n // 64)
        jobs = [(name, lo, min(lo + step, n)) for lo in range(0, n, step)]
        with get_context("fork").Pool(16) as pool:
            ref = np.concatenate(pool.map(work, jobs))
        tol = TOL[name]
        bad = parity.compare(got, ref, COLUMNS[name], int_cols=INT.get(name, ()), label=name, **tol)
        print(name, n, "objects:", len(bad), "mismatches", f"({time.time() - t0:.1f}s)", flush=True)
        for b in bad[:8]:
            print("   ", b)
