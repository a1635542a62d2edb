"""Parity at the BASELINE.json configuration sizes and seeds, through the C-ABI on an MI355X.

C1   3,054 objects (seed 3054): the statistics set against the oracle, every object.
C2-4 10,178 objects (seed 10178): all eight v34a/v55 sets, every 100th object against the oracle
     (test_gpu_parity.py::test_full_size_properties holds the size-independent properties).
C5   the 1 M-object survey at FULL size: its eight 125,000-object blocks (seeds 1000000 .. 1000007, exactly the blocks
     bench.py's survey is made of), all eight sets; per block shape, integer counts and an oracle sample (every 400th
     object of block 0, every 2,000th of the others); block 0 twice for run-to-run identity.

Rules.  Streaming sets (stat, tde, color, shape, physics): identical NaN mask, bit-exact integer columns,
floats within the tolerances of test_gpu_parity.TOL.  Bounded fits and the GP (no probe runs exist for
seeded samples, so the stability-aware rule of conftest.check_fit_parity cannot be applied): NaN-mask
mismatches <= 1 % of the entries and the share of values within 1e-4 relative at least

    bazin 0.78   powerlaw 0.97   gp2d 0.95

(measured on 1,500 / 400 objects with tools/parity_sweep.py: 0.834 / 0.990 / 0.987 -- the thresholds sit 5 points
below; the reference's own self-agreement under one-ulp probes on the golden set is 0.62 for the Bazin parameters,
DESIGN.md §5).  The shares are taken over the pooled sample of a test (>= 100 objects), not per block.
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
FIT_MIN_CLOSE = {"bazin": 0.78, "powerlaw": 0.97, "gp2d": 0.95}


def sample_counts(got_all, lc, rows):
    """Compare rows `rows` of a multi-set result with the oracle run on exactly those objects.  Streaming sets are
    asserted here; for the fit sets the counts (values within 1e-4, values compared, NaN-mask mismatches, entries)
    are returned so that a test can pool them over its samples."""
    sub = synth_subset.take(lc, rows)
    col0 = 0
    counts = {}
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
        both = ~np.isnan(got) & ~np.isnan(ref)
        with np.errstate(all="ignore"):
            rel = np.abs(got - ref)[both] / np.maximum(np.abs(ref[both]), 1e-9)
        counts[name] = np.array([int((rel <= 1e-4).sum()), int(both.sum()), int((np.isnan(got) != np.isnan(ref)).sum()), got.size])
    return counts


def assert_fit_shares(counts):
    report = {}
    for name, (close, n, nan_mis, size) in counts.items():
        report[name] = (close / max(n, 1), nan_mis / size)
        assert nan_mis <= 0.01 * size, (name, nan_mis, size)
        assert close >= FIT_MIN_CLOSE[name] * n, (name, close / max(n, 1))
    print("fit sets: share within 1e-4, NaN-mask mismatch share:", report)


def check_sample(got_all, lc, rows):
    assert_fit_shares(sample_counts(got_all, lc, rows))


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


def test_config5_full_survey():
    """Config 5 at full size: the 1 M-object survey bench.py draws (block b = make_lightcurves(125000, 1000000 + b))."""
    n = 125000
    key = lambda m: np.nan_to_num(m, nan=-7.25e300)
    cols = COLUMNS["stat"]
    nobs_cols = [cols.index(f"{p}_n_obs") for p in "ugrizy"]
    pooled = {}
    total = 0
    for b in range(8):
        lc = synth.make_lightcurves(n, seed=1000000 + b)
        a = extract_csr(SETS, lc, z=lc["z"])
        assert a.shape == (n, 434)
        if b == 0:
            again = extract_csr(SETS, lc, z=lc["z"])
            assert np.array_equal(key(a), key(again)), "two runs on the same block differ"
            del again
        # counts are integers and add up at full size
        nobs = a[:, nobs_cols]
        assert np.array_equal(nobs, np.round(nobs)) and np.array_equal(nobs.sum(1), np.diff(lc["offsets"]).astype(float)), b
        assert np.array_equal(a[:, cols.index("all_n_obs")], np.diff(lc["offsets"]).astype(float)), b
        for name, c in sample_counts(a, lc, list(range(0, n, 400 if b == 0 else 2000))).items():
            pooled[name] = pooled.get(name, 0) + c
        total += n
        print(f"block {b}: {n} objects, {int(lc['offsets'][-1])} points ok", flush=True)
        del a, lc
    assert total == 1000000
    assert_fit_shares(pooled)
