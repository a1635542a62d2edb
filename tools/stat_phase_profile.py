"""Debug helper: cycle shares of the statistics kernel's phases (library built with -DLCFE_PHASE_PROF,
loaded through LCFE_LIB_PATH)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mallorn_astrophysics_amd import synth, _lib
from mallorn_astrophysics_amd.engine import extract_csr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
lc = synth.make_lightcurves(n, seed=5)
lib = _lib.load()
buf = (ctypes.c_ulonglong * 32)()
extract_csr("stat", lc)
lib.lcfe_debug_phase_prof(buf)
extract_csr("stat", lc)
lib.lcfe_debug_phase_prof(buf)
names = {0: "stage", 1: "six bands (8-lane groups)", 2: "all rows (wave)", 3: "store row", 4: "ratios/peak_band", 5: "ticket + offsets + skipped objects"}
sub = ["pass 1 / combine", "pass 2", "pass 3", "sort", "ranks + MAD", "slope", "epilogue"]
for k in range(7):
    names[8 + k] = "  all: " + sub[k]
    names[16 + k] = "  band group 0: " + sub[k]
for k, nm in ((8, "median"), (9, "iqr"), (10, "mad")):
    names[16 + k] = "  all: " + nm
    names[24 + k] = "  band group 0: " + nm
tot = sum(buf[k] for k in range(6))
for k in sorted(names):
    if k < 32:
        print(f"{names[k]:34s} {buf[k] / n:9.0f} cycles/object {100 * buf[k] / tot:5.1f}%")
print("total cycles/object", tot / n)
