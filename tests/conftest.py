import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_inputs():
    g = np.load(os.path.join(GOLDEN, "golden_inputs.npz"))
    return {k: g[k] for k in g.files}


def load_golden(name):
    return np.load(os.path.join(GOLDEN, f"golden_{name}.npz"))["out"]


def load_golden_fit(name):
    """(ref, probes) for the bounded-fit sets: probes = the reference re-run under one-ulp flux
    perturbations (out_p*) and one-ulp model-evaluation noise (out_m*)."""
    g = np.load(os.path.join(GOLDEN, f"golden_{name}.npz"))
    return g["out"], [g[k] for k in ("out_p1", "out_p2", "out_m1", "out_m2", "out_m3")]


def check_fit_parity(got, name, cols, max_stable_bad=2, ref=None, probes=None):
    """Parity rule of the bounded fits (DESIGN.md "Parity of the bounded fits")."""
    import parity
    if ref is None:
        ref, probes = load_golden_fit(name)
    bad, summ = parity.compare_fits(got, ref, probes, name, cols, rtol=1e-4)
    print(name, summ)
    # (1) fits the reference reproduces under one-ulp probes must match to 1e-4 (a handful of
    #     kink-crossing exceptions in degenerate valleys are tolerated and listed)
    assert len(bad) <= max_stable_bad, "\n".join(bad)
    # (2) over ALL fits the implementation is as close to the reference as the reference is to
    #     itself under one-ulp noise
    assert summ["close_frac"] >= summ["scipy_self_close_frac"] - 0.05, summ
    # (3) NaN mask: only chaotic fits (max_nfev / infeasible after a long walk) may differ
    assert summ["nan_mask_mismatches"] <= 0.01 * summ["n_fits"], summ
    return summ


def load_gp_oracle_fixture():
    """ORACLE outputs (parity unpinned: george absent) + two one-ulp probe runs."""
    g = np.load(os.path.join(GOLDEN, "golden_gp2d_oracle.npz"))
    return g["out"], [g["out_p1"], g["out_p2"]]
