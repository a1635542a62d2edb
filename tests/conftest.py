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


def check_cost_parity(got, status, name, csr):
    """Parity of what the bounded fits actually converge -- the cost -- and of the evaluation counts.

    scipy's TRF stops at ftol = xtol = gtol = 1e-8 on the COST (bazin_fitting.py:128-137,
    train_v55_powerlaw.py:173-184), so the cost, unlike the parameters of a flat valley, is pinned:
      (1) the share of fits whose cost (reduced chi^2 / R^2) is within 1e-6 relative of the reference's must reach
          the reference's own share under one-ulp probes, minus 2 points;
      (2) the share of fits that end more than 1e-3 (relative) WORSE than the reference may exceed the reference's
          own rate by at most 1 point;
      (3) Bazin: on the fits the reference reproduces under every probe, nfev equals golden_bazin.npz['nfev'] for
          at least 95 %, and a fit the reference completed is never reported as failed there."""
    import parity
    ref, probes = load_golden_fit(name)
    summ = parity.compare_cost(got, ref, probes, name)
    print(name, "cost parity", summ)
    assert summ["close_1e-6"] >= summ["self_close_1e-6"] - 0.02, summ
    assert summ["worse_1e-3"] <= summ["self_worse_1e-3"] + 0.01, summ
    if name == "bazin":
        tab = parity.bazin_nfev_table(csr, np.load(os.path.join(GOLDEN, "golden_bazin.npz"))["nfev"])
        stable = parity.fit_stability(ref, probes, parity.fit_blocks("bazin"))
        st, nf = status[:, 0::2], status[:, 1::2]
        done = (tab > 0) & stable
        assert (st[done] > 0).all(), "a stable fit the reference completed is reported as failed"
        eq = float((nf[done] == tab[done]).mean())
        print("bazin nfev equal on", int(done.sum()), "stable fits:", eq)
        assert eq >= 0.95, eq
    return summ
