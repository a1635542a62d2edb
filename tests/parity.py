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


def fit_blocks(name):
    """Column blocks that come from ONE bounded fit: (start, stop) pairs."""
    if name == "bazin":
        return [(8 * k, 8 * k + 8) for k in range(6)]
    if name == "powerlaw":
        return [(j, j + 1) for j in range(27)]
    if name == "gp2d":
        return [(0, 27)]
    if name == "gp1d":
        # four columns per band fit (g, r, i, z), then the five cross-band columns derived from all of them
        return [(4 * k, 4 * k + 4) for k in range(4)] + [(16, 21)]
    raise KeyError(name)


def _rel(a, b, floor):
    with np.errstate(all="ignore"):
        r = np.abs(a - b) / np.maximum(np.abs(b), floor)
    return r


def fit_stability(ref, probes, blocks, tol=1e-6, floor=1e-9):
    """Per (object, block): True where the REFERENCE reproduces itself under every probe run
    (one-ulp flux perturbations, one-ulp model-evaluation noise): same NaN pattern and values
    within `tol`.  See tests/golden/make_golden.py for how the probes are made."""
    n = len(ref)
    stable = np.ones((n, len(blocks)), bool)
    for j, (a, b) in enumerate(blocks):
        r = ref[:, a:b]
        for pr in probes:
            x = pr[:, a:b]
            nan_same = (np.isnan(r) == np.isnan(x)).all(1)
            d = _rel(x, r, floor)
            d[np.isnan(r) & np.isnan(x)] = 0
            d[np.isnan(d)] = np.inf
            stable[:, j] &= nan_same & (d.max(1) < tol)
    return stable


def compare_fits(got, ref, probes, name, cols, rtol=1e-4, floor=1e-9):
    """Parity of bounded-fit feature sets.

    Fits the reference itself reproduces under a one-ulp perturbation ("stable") must match in NaN
    mask and within `rtol`.  For the others only distribution-level agreement is meaningful: the
    caller gets the fraction of all fits within `rtol` for the implementation and for
    scipy-vs-perturbed-scipy."""
    blocks = fit_blocks(name)
    stable = fit_stability(ref, probes, blocks)
    p1 = probes[-1]
    bad = []
    close_got, close_self, nan_mis = [], [], 0
    for j, (a, b) in enumerate(blocks):
        g, r = got[:, a:b], ref[:, a:b]
        nan_eq = (np.isnan(g) == np.isnan(r)).all(1)
        d = _rel(g, r, floor)
        d[np.isnan(g) & np.isnan(r)] = 0
        d[np.isnan(d)] = np.inf
        ok = nan_eq & (d.max(1) <= rtol)
        ds = _rel(p1[:, a:b], r, floor)
        ds[np.isnan(p1[:, a:b]) & np.isnan(r)] = 0
        ds[np.isnan(ds)] = np.inf
        close_got.append(ok)
        close_self.append(ds.max(1) <= rtol)
        nan_mis += int((~nan_eq).sum())
        for i in np.flatnonzero(stable[:, j] & ~ok)[:5]:
            bad.append(f"{name} stable fit obj {i} cols {cols[a]}..: got {g[i]} ref {r[i]}")
    close_got, close_self = np.array(close_got), np.array(close_self)
    attempted = ~np.isnan(np.stack([ref[:, a] for a, _ in blocks])) | ~np.isnan(np.stack([got[:, a] for a, _ in blocks]))
    summary = {"n_fits": int(attempted.sum()), "stable_frac": float(stable.T[attempted].mean()),
               "close_frac": float(close_got[attempted].mean()),
               "scipy_self_close_frac": float(close_self[attempted].mean()),
               "nan_mask_mismatches": nan_mis}
    return bad, summary


def cost_columns(name):
    """Index of the converged-cost column of every fit block: the reduced chi^2 of a Bazin band fit
    (bazin_fitting.py:148-151) / the R^2 of a decline model (train_v55_powerlaw.py:186-190)."""
    if name == "bazin":
        return [8 * k + 5 for k in range(6)], +1          # lower is better
    if name == "powerlaw":
        return list(range(27)), -1                        # higher is better
    raise KeyError(name)


def compare_cost(got, ref, probes, name, floor=1e-12):
    """What TRF actually converges is the COST, to ftol = 1e-8.  Returns the shares, over the fits both sides
    carried out, of (a) costs within 1e-6 relative of the reference's and (b) costs worse than the reference's by
    more than 1e-3 relative, for the implementation and for the reference re-run under one-ulp probes."""
    cols, sign = cost_columns(name)
    g, r = got[:, cols], ref[:, cols]
    both = ~np.isnan(g) & ~np.isnan(r)

    def shares(x, mask):
        with np.errstate(all="ignore"):
            d = (x - r) / np.maximum(np.abs(r), floor)
        d = d[mask]
        return float((np.abs(d) <= 1e-6).mean()), float((sign * d > 1e-3).mean())

    close, worse = shares(g, both)
    self_close, self_worse = [], []
    for p in probes:
        x = p[:, cols]
        c, w = shares(x, ~np.isnan(x) & ~np.isnan(r))
        self_close.append(c)
        self_worse.append(w)
    return {"n": int(both.sum()), "close_1e-6": close, "worse_1e-3": worse,
            "self_close_1e-6": float(np.mean(self_close)), "self_worse_1e-3": float(np.mean(self_worse))}


def bazin_nfev_table(csr, nfev_log):
    """[n_obj, 6] table of the reference's nfev per band fit from the call-order log of
    tests/golden/make_golden.py (one entry per band with >= 5 rows, -1 where curve_fit raised); -2 = no call."""
    off, band = csr["offsets"], csr["band"]
    tab = np.full((len(off) - 1, 6), -2, np.int64)
    k = 0
    for i in range(len(off) - 1):
        b = band[off[i]:off[i + 1]]
        for j in range(6):
            if (b == j).sum() >= 5:
                tab[i, j] = nfev_log[k]
                k += 1
    assert k == len(nfev_log)
    return tab
