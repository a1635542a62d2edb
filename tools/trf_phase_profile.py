"""Debug helper: cycle shares of the TRF phases (library built with -DLCFE_TRF_PROF via LCFE_LIB_PATH)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mallorn_astrophysics_amd import synth, _lib
from mallorn_astrophysics_amd.engine import extract_csr

lc = synth.make_lightcurves(int(sys.argv[1]) if len(sys.argv) > 1 else 4000, seed=5)
lib = _lib.load()
buf = (ctypes.c_ulonglong * 8)()
out, st = extract_csr("bazin", lc, return_status=True)
lib.lcfe_debug_trf_prof(buf)
nfev = st[:, 1::2].sum()
names = ["CL scale + augment", "Householder QR", "Jacobi SVD", "TR solve", "select step", "residual", "accept/copy", "FD jacobian"]
tot = sum(buf)
for n, v in zip(names, buf):
    print(f"{n:20s} {v / nfev:10.0f} cycles/nfev {100 * v / tot:5.1f}%")
print("total cycles/nfev", tot / nfev, "nfev", nfev)
