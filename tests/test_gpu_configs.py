"""Parity at the BASELINE.json configuration sizes and seeds, through the C-ABI on an MI355X.

C1   3,054 objects (seed 3054): the statistics set against the oracle, every object.
C2-4 10,178 objects (seed 10178): all eight v34a/v55 sets, every 100th object against the oracle
     (test_gpu_parity.py::test_full_size_properties holds the size-independent properties).
C5   one 125,000-object shard (seed 1000000) of the 1 M-object survey: run-to-run identity and every 400th
     object of all eight sets against the oracle.

Rules.  Streaming sets (stat, tde, color, shape, physics): identical NaN mask, bit-exact integer columns,
floats within the tolerances of test_gpu_parity.TOL.  Bounded fits and the GP (no probe runs exist for
seeded samples, so the stability-aware rule of conftest.check_fit_parity cannot be applied): NaN-mask
mismatches <= 1 % of the entries and the share of values within 1e-4 relative at least

    bazin 0.75   powerlaw 0.95   gp2d 0.93

(measured on 1,500 / 400 objects with tools/parity_sweep.py: 0.834 / 0.990 / 0.987; the reference's own
self-agreement under one-ulp probes on the golden set is 0.62 for the Bazin parameters).
"""
import numpy as np
import pytest

import oracle
import parity
import synth_subset
from mallorn_astrophysics_amd import synth
from mallorn_astrophysics_amd.columns import COLUMNS, STAT_INT_COLUMNS
from mallorn_astrophysics_amd.engine import extract_csr
from test_gpu_parity import TOL

pytestmark = pytest.mark.gpu

SETS = ["stat", "bazin", "powerlaw", "tde", "color", "shape", "physics", "gp2d"]
FIT_MIN_CLOSE = {"bazin": 0.75, "powerlaw": 0.95, "gp2d": 0.93}


def check_sample(got_all, lc, rows):
    """Compare rows `rows` of a multi-set result with the oracle run on exactly those objects."""
    sub = synth_subset.take(lc, rows)
    col0 = 0
    report = {}
    for name in SETS:
        ncol = len(COLUMNS[name])
        got = got_all[rows, col0:col0 + ncol]
        col0 += ncol
        ref = oracle.extract(name, sub, sub["z"])
        if name in TOL:
            bad = parity.compare(got, ref, COLUMNS[name], int_cols=STAT_INT_COLUMNS if name == "stat" else (),
                                 label=name, **TOL[name])
            assert not bad, "\n".join(bad)
            continue
        nan_mis = float((np.isnan(got) != np.isnan(ref)).mean())
        both = ~np.isnan(got) & ~np.isnan(ref)
        with np.errstate(all="ignore"):
            rel = np.abs(got - ref)[both] / np.maximum(np.abs(ref[both]), 1e-9)
        close = float((rel <= 1e-4).mean())
        report[name] = (close, nan_mis)
        assert nan_mis <= 0.01, (name, nan_mis)
        assert close >= FIT_MIN_CLOSE[name], (name, close)
    print("fit sets: share within 1e-4, NaN-mask mismatch share:", report)


def test_config1_statistics_3054_objects_full_oracle():
    lc = synth.make_lightcurves(3054, seed=3054)
    got = extract_csr("stat", lc)
    ref = oracle.extract("stat", lc)
    bad = parity.compare(got, ref, COLUMNS["stat"], int_cols=STAT_INT_COLUMNS, label="stat", **TOL["stat"])
    assert not bad, "\n".join(bad)


def test_config2to4_10178_objects_oracle_sample():
    lc = synth.make_lightcurves(10178, seed=10178)
    got = extract_csr(SETS, lc, z=lc["z"])
    assert got.shape == (10178, 434)
    check_sample(got, lc, list(range(0, 10178, 100)))


def test_config5_shard_125000_objects_identity_and_oracle_sample():
    n = 125000
    lc = synth.make_lightcurves(n, seed=1000000)
    a = extract_csr(SETS, lc, z=lc["z"])
    b = extract_csr(SETS, lc, z=lc["z"])
    key = lambda m: np.nan_to_num(m, nan=-7.25e300)
    assert np.array_equal(key(a), key(b)), "two runs on the same shard differ"
    del b
    # counts are integers and add up at full size
    cols = COLUMNS["stat"]
    nobs = a[:, [cols.index(f"{p}_n_obs") for p in "ugrizy"]]
    assert np.array_equal(nobs.sum(1), np.diff(lc["offsets"]).astype(float))
    check_sample(a, lc, list(range(0, n, 400)))
