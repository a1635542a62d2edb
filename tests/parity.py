"""Parity rules shared by the tests: identical NaN mask, bit-exact integer
columns, floats within a stated relative tolerance."""
import numpy as np


def compare(got, ref, cols, rtol, atol=0.0, int_cols=(), skip_rows=(), label=""):
    """Return a list of human-readable mismatches (empty = parity)."""
    got = np.asarray(got, float)
    ref = np.asarray(ref, float)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    keep = np.ones(len(ref), bool)
    keep[list(skip_rows)] = False
    bad = []
    nan_mis = (np.isnan(got) != np.isnan(ref)) & keep[:, None]
    for i, j in np.argwhere(nan_mis)[:10]:
        bad.append(f"{label} NaN-mask obj {i} col {cols[j]}: got {got[i, j]!r} ref {ref[i, j]!r}")
    both = ~np.isnan(got) & ~np.isnan(ref) & keep[:, None]
    with np.errstate(all="ignore"):
        err = np.abs(got - ref)
        tol = atol + rtol * np.abs(ref)
        inf_ok = np.isinf(got) & np.isinf(ref) & (np.sign(got) == np.sign(ref))
        viol = both & ~(err <= tol) & ~inf_ok
    for j, c in enumerate(cols):
        if c in int_cols:
            viol[:, j] = both[:, j] & (got[:, j] != ref[:, j])
    for i, j in np.argwhere(viol)[:10]:
        bad.append(f"{label} value obj {i} col {cols[j]}: got {got[i, j]!r} ref {ref[i, j]!r}")
    if nan_mis.sum() + viol.sum() > len(bad):
        bad.append(f"{label} ... {int(nan_mis.sum())} NaN-mask and {int(viol.sum())} value mismatches in total")
    return bad


def max_rel(got, ref):
    with np.errstate(all="ignore"):
        r = np.abs(got - ref) / np.abs(ref)
    r[~np.isfinite(r)] = 0
    r[got == ref] = 0
    return float(np.nanmax(r)) if r.size else 0.0
