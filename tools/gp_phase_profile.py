"""Debug helper: per-phase cycle shares of the GP kernel (needs a library built with -DLCFE_GP_PROF,
passed through LCFE_LIB_PATH).  Prints cycles (x1024) per phase summed over objects, by N bucket."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mallorn_astrophysics_amd import synth
from mallorn_astrophysics_amd.engine import extract_csr

lc = synth.make_lightcurves(int(sys.argv[1]) if len(sys.argv) > 1 else 4000, seed=5)
out, st = extract_csr("gp2d", lc, return_status=True)
n = st[:, 3]
names = ["V gather", "P inverse", "Wm", "fixup+barrier", "gram", "sweep total", "alpha", "grad", "tile update", "optimiser"]
for lo, hi in ((0, 63), (63, 111), (111, 159), (159, 300), (300, 800)):
    m = (n > lo) & (n <= hi)
    if not m.any():
        continue
    tot = st[m, 4:14].sum(0).astype(float)
    ev = st[m, 2].sum()
    print(f"N in ({lo},{hi}]: {m.sum()} objects, {ev} evals, mean N {n[m].mean():.0f}")
    denom = tot[4] + tot[5] + tot[6] + tot[7]
    for k, nm in enumerate(names):
        print(f"   {nm:12s} {tot[k] / ev * 1024:12.0f} cycles/eval  {100 * tot[k] / denom:5.1f}%")
