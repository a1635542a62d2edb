"""Debug helper: run one feature set with two builds of liblcfe.so and compare the outputs bit for bit.
usage: compare_libs.py SET N_OBJ SEED LIB_A LIB_B"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 6:
    name, n, seed, a, b = sys.argv[1:]
    outs = []
    for lib in (a, b):
        path = f"/tmp/cmp_{os.path.basename(lib)}.npy"
        env = dict(os.environ, LCFE_LIB_PATH=os.path.abspath(lib))
        subprocess.run([sys.executable, __file__, name, n, seed, path], env=env, check=True)
        outs.append(np.load(path))
    x, y = outs
    same = (x == y) | (np.isnan(x) & np.isnan(y))
    print("identical:", bool(same.all()), "differing entries:", int((~same).sum()), "of", same.size)
    if not same.all():
        with np.errstate(all="ignore"):
            rel = np.abs(x - y) / np.maximum(np.abs(y), 1e-300)
        rel[same] = 0
        print("max rel diff", np.nanmax(rel), "objects differing", int((~same).any(1).sum()))
    sys.exit(0 if same.all() else 1)
else:
    sys.path.insert(0, ROOT)
    from mallorn_astrophysics_amd import synth
    from mallorn_astrophysics_amd.engine import extract_csr
    name, n, seed, path = sys.argv[1:]
    lc = synth.make_lightcurves(int(n), seed=int(seed))
    np.save(path, extract_csr(name, lc, z=lc["z"]))
