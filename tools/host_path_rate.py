import sys, time
sys.path.insert(0, '.')
import numpy as np
from mallorn_astrophysics_amd import synth
from mallorn_astrophysics_amd.engine import extract_csr
from mallorn_astrophysics_amd.columns import SET_NAMES
lc = synth.make_lightcurves(125000, seed=1000000)
sets = list(SET_NAMES)
extract_csr(["stat"], lc, z=lc["z"])
for trial in range(2):
    t0 = time.perf_counter()
    out, prof = extract_csr(sets, lc, z=lc["z"], return_prof=True)
    dt = time.perf_counter() - t0
    print("host-buffer call: %.3f s -> %.0f light curves/s; h2d %.1f ms, d2h %.1f ms, bytes_in %.0f MB, bytes_out %.0f MB" % (
        dt, 125000 / dt, prof["h2d_ms"], prof["d2h_ms"], prof["bytes_in"] / 1e6, prof["bytes_out"] / 1e6))
    t0 = time.perf_counter()
    out, prof = extract_csr(["stat"], lc, return_prof=True)
    dt = time.perf_counter() - t0
    print("stat only host-buffer call: %.4f s; h2d %.1f ms, d2h %.1f ms kernel %.2f ms" % (dt, prof["h2d_ms"], prof["d2h_ms"], prof["kernel_ms"][0]))
