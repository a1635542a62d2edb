"""Deterministic synthetic LSST-like 6-band light curves.

There is no competition data in the build or bench environment, so every test, fixture and
benchmark draws its light curves from this generator.  The output is already CSR-packed (the
layout the device consumes); :func:`to_dataframe` turns it into the long DataFrame the
reference's ``extract_*_features`` functions take (columns as in
``src/utils/data_loader.py:36-62`` / ``src/features/statistical.py:144`` of the reference).
"""
from __future__ import annotations

import numpy as np

BANDS = "ugrizy"
BAND_PROB = np.array([0.07, 0.12, 0.25, 0.25, 0.19, 0.12])
BAND_SIGMA = np.array([2.5, 1.0, 0.8, 0.9, 1.5, 3.0])
BAND_WAVE_A = np.array([3670.0, 4825.0, 6222.0, 7545.0, 8691.0, 9710.0])
T_LO, T_HI = 59000.0, 60000.0


def _bazin(t, A, t0, tr, tf):
    x = t - t0
    with np.errstate(over="ignore", invalid="ignore"):
        v = A * np.exp(-x / tf) / (1.0 + np.exp(-x / tr))
    return np.nan_to_num(v, nan=0.0, posinf=0.0, neginf=0.0)


def make_lightcurves(n_obj: int, seed: int | None = None, n_min: int = 12, n_max: int = 500,
                     n_median: float = 120.0, t_span: float = T_HI - T_LO):
    """Return a dict of CSR arrays for ``n_obj`` synthetic objects.

    keys: ``offsets`` int64[n_obj+1], ``t``/``flux``/``err`` float64[total], ``band`` uint8[total]
    (0..5 = u,g,r,i,z,y), ``z``/``ebv`` float64[n_obj], ``cls`` uint8[n_obj].
    Rows of one object are stored in time order (as the competition CSVs are).
    ``t_span`` (default 1000 d, the SURVEY §8d spec) shortens the observing window for the
    denser fixture objects.
    """
    rng = np.random.default_rng(n_obj if seed is None else seed)
    n = np.clip(np.rint(rng.lognormal(np.log(n_median), 0.5, n_obj)), n_min, n_max).astype(np.int64)
    offsets = np.zeros(n_obj + 1, np.int64)
    np.cumsum(n, out=offsets[1:])
    total = int(offsets[-1])
    obj = np.repeat(np.arange(n_obj), n)

    t = rng.uniform(T_LO, T_LO + t_span, total)
    order = np.lexsort((t, obj))
    t = t[order]
    # exact duplicate times inside an object are re-drawn (vanishingly rare with doubles)
    dup = np.flatnonzero((np.diff(t) == 0) & (np.diff(obj) == 0))
    while dup.size:
        t[dup + 1] = np.nextafter(t[dup + 1], np.inf)
        dup = np.flatnonzero((np.diff(t) == 0) & (np.diff(obj) == 0))
    band = rng.choice(6, size=total, p=BAND_PROB).astype(np.uint8)

    cls = rng.choice(4, size=n_obj, p=[0.4, 0.1, 0.3, 0.2]).astype(np.uint8)  # bazin,tde,agn,noise
    A = rng.lognormal(np.log(40.0), 1.0, n_obj)
    tr = rng.uniform(2.0, 20.0, n_obj)
    tf = rng.uniform(10.0, 120.0, n_obj)
    t0 = rng.uniform(T_LO + 0.15 * t_span, T_LO + 0.85 * t_span, n_obj)
    beta = rng.normal(1.0, 0.7, n_obj)
    base = rng.normal(0.0, 1.0, n_obj)
    tau_drw = rng.uniform(50.0, 500.0, n_obj)
    sig_drw = rng.uniform(1.0, 10.0, n_obj)
    mean_drw = rng.uniform(5.0, 50.0, n_obj)
    z = rng.uniform(0.01, 1.2, n_obj)
    ebv = rng.uniform(0.0, 0.3, n_obj)

    c = cls[obj]
    amp_b = (BAND_WAVE_A[2] / BAND_WAVE_A[band]) ** beta[obj]
    model = np.zeros(total)
    m = c == 0
    model[m] = amp_b[m] * _bazin(t[m], A[obj[m]], t0[obj[m]], tr[obj[m]], tf[obj[m]]) + base[obj[m]]
    m = c == 1
    if m.any():
        o = obj[m]
        x = t[m] - t0[o]
        rise = _bazin(t[m], A[o], t0[o], tr[o], 1e9)
        decay = np.where(x > 0, (1.0 + np.maximum(x, 0) / tf[o]) ** (-5.0 / 3.0), 1.0)
        model[m] = amp_b[m] * rise * decay + base[o]
    # damped random walk: sequential in the point index, vectorised across AGN objects
    agn = np.flatnonzero(cls == 2)
    if agn.size:
        eps = rng.standard_normal(total)
        x = sig_drw[agn] * eps[offsets[agn]]
        model[offsets[agn]] = mean_drw[agn] + x
        alive = np.arange(agn.size)
        for k in range(1, int(n[agn].max())):
            alive = alive[n[agn[alive]] > k]
            if alive.size == 0:
                break
            a = agn[alive]
            idx = offsets[a] + k
            r = np.exp(-(t[idx] - t[idx - 1]) / tau_drw[a])
            x_prev = model[idx - 1] - mean_drw[a]
            xk = x_prev * r + sig_drw[a] * np.sqrt(1.0 - r * r) * eps[idx]
            model[idx] = mean_drw[a] + xk
    sb = BAND_SIGMA[band]
    err = np.abs(rng.normal(sb, 0.2 * sb)) + 0.05
    flux = model + rng.normal(0.0, 1.0, total) * err
    return {"offsets": offsets, "t": t, "flux": flux, "err": err, "band": band,
            "z": z, "ebv": ebv, "cls": cls}


def lengths(n_obj: int, seed: int | None = None, n_min: int = 12, n_max: int = 500, n_median: float = 120.0):
    """Row counts of the objects :func:`make_lightcurves` draws for the same arguments (its first random draw),
    without generating the light curves: lets a rank of a sharded run locate its objects in a large survey."""
    rng = np.random.default_rng(n_obj if seed is None else seed)
    return np.clip(np.rint(rng.lognormal(np.log(n_median), 0.5, n_obj)), n_min, n_max).astype(np.int64)


def slice_objects(lc: dict, lo: int, hi: int):
    """Objects [lo, hi) of a CSR dict as a CSR dict of their own (contiguous slice, no copy of the other rows)."""
    off = np.asarray(lc["offsets"], np.int64)
    s, e = int(off[lo]), int(off[hi])
    out = {"offsets": np.ascontiguousarray(off[lo:hi + 1] - s)}
    for k in ("t", "flux", "err", "band"):
        out[k] = np.ascontiguousarray(lc[k][s:e])
    for k in ("z", "ebv", "cls"):
        if k in lc:
            out[k] = np.ascontiguousarray(np.asarray(lc[k])[lo:hi])
    return out


def object_ids(n_obj: int, prefix: str = "obj"):
    return [f"{prefix}_{i:07d}" for i in range(n_obj)]


def to_dataframe(lc: dict, ids=None):
    """Long DataFrame (object_id, Time (MJD), Flux, Flux_err, Filter) + metadata frame."""
    import pandas as pd

    n_obj = len(lc["offsets"]) - 1
    ids = object_ids(n_obj) if ids is None else list(ids)
    n = np.diff(lc["offsets"])
    band = np.asarray(lc["band"])
    filt = np.array(list(BANDS) + ["?"], dtype=object)[np.where(band < 6, band, 6)]
    df = pd.DataFrame({
        "object_id": np.repeat(np.array(ids, dtype=object), n),
        "Time (MJD)": lc["t"], "Flux": lc["flux"], "Flux_err": lc["err"], "Filter": filt,
    })
    meta = pd.DataFrame({"object_id": ids, "Z": lc.get("z", np.full(n_obj, np.nan)),
                         "EBV": lc.get("ebv", np.full(n_obj, np.nan))})
    return df, meta


def concat(parts):
    """Concatenate CSR dicts (used to append hand-written edge cases to a random set)."""
    out = {k: np.concatenate([p[k] for p in parts]) for k in ("t", "flux", "err", "band", "z", "ebv")}
    offs = [np.zeros(1, np.int64)]
    base = 0
    for p in parts:
        offs.append(p["offsets"][1:] + base)
        base += int(p["offsets"][-1])
    out["offsets"] = np.concatenate(offs)
    return out


def from_objects(objs):
    """Build a CSR dict from a list of (t, flux, err, band[, z]) tuples (band as str or ints)."""
    ts, fs, es, bs, zs, offs = [], [], [], [], [], [0]
    for o in objs:
        t, f, e, b = o[:4]
        zz = o[4] if len(o) > 4 else 0.1
        t = np.asarray(t, float)
        if isinstance(b, str):
            b = [BANDS.index(ch) if ch in BANDS else 255 for ch in b]
        b = np.asarray(b, np.uint8)
        if b.size == 1 and t.size != 1:
            b = np.full(t.size, b[0], np.uint8)
        ts.append(t); fs.append(np.asarray(f, float)); es.append(np.asarray(e, float)); bs.append(b)
        zs.append(zz)
        offs.append(offs[-1] + t.size)
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
    return {"offsets": np.asarray(offs, np.int64), "t": cat(ts, float), "flux": cat(fs, float),
            "err": cat(es, float), "band": cat(bs, np.uint8), "z": np.asarray(zs, float),
            "ebv": np.zeros(len(objs))}


def edge_cases():
    """Hand-written objects covering the edge cases SURVEY.md §8d / Appendix C lists."""
    rng = np.random.default_rng(99)
    T = 59000.0
    kt = T + np.array([0, 3, 7, 12, 18, 25, 33, 42, 52, 63, 75, 88.0])
    kf = np.array([1, 2.5, 9, 21, 30, 26, 20.5, 15, 11.5, 8, 6.5, 5])
    ke = np.array([.5, .6, .7, .8, .9, 1, .9, .8, .7, .6, .5, .5])
    objs = []
    # 0: the known-answer vector of SURVEY §8c, r band only
    objs.append((kt, kf, ke, "r" * 12))
    # 1: same curve in g, r, i with offsets (colour features defined)
    objs.append((np.concatenate([kt, kt + 0.3, kt + 0.6]), np.concatenate([kf * 1.3, kf, kf * 0.8]),
                 np.concatenate([ke, ke, ke]), "g" * 12 + "r" * 12 + "i" * 12))
    # 2: single point
    objs.append(([T + 1.0], [3.0], [0.5], "r"))
    # 3: bands with 1, 2, 4, 5 points
    t = T + np.arange(12) * 2.5
    objs.append((t, rng.normal(5, 2, 12), np.full(12, 0.7), "u" + "gg" + "rrrr" + "iiiii"))
    # 4: all fluxes negative
    objs.append((T + np.arange(8) * 4.0, -np.abs(rng.normal(5, 1, 8)) - 0.1, np.full(8, 0.5), "r" * 8))
    # 5: constant flux
    objs.append((T + np.array([0, 3, 7, 12, 18, 25, 33, 42.0]), np.full(8, 5.0), np.full(8, 0.5), "r" * 8))
    # 6: duplicate times (documented, not gated)
    objs.append((T + np.array([0, 1, 1, 2, 3, 5, 8, 8, 13.0]), rng.normal(10, 3, 9), np.full(9, 1.0), "g" * 9))
    # 7: err <= 0 and NaN err
    e = np.full(10, 0.8); e[2] = 0.0; e[5] = -1.0; e[7] = np.nan
    objs.append((T + np.arange(10) * 3.0, rng.normal(8, 3, 10), e, "r" * 10))
    # 8: N < 10 spread over bands
    objs.append((T + np.arange(7) * 5.0, rng.normal(4, 1, 7), np.full(7, 0.6), "ugrizyr"))
    # 9: rows not in time order (file order != time order), 3 bands
    t = T + rng.permutation(40) * 2.0 + rng.uniform(0, 1, 40)
    f = 20 * np.exp(-0.5 * ((t - T - 30) / 12) ** 2) + rng.normal(0, 1, 40)
    objs.append((t, f, np.full(40, 1.0), "".join(rng.choice(list("gri"), 40))))
    # 10: short duration (0.4 d) -> infeasible Bazin start
    objs.append((T + np.linspace(0, 0.4, 6), rng.normal(5, 1, 6) + 5, np.full(6, 0.5), "r" * 6))
    # 11: exactly 5 points in one band
    objs.append((T + np.array([0, 5, 11, 18, 30.0]), np.array([2.0, 9, 14, 8, 3]), np.full(5, 0.7), "i" * 5))
    # 12: median < -f_max (B0 below its lower bound)
    objs.append((T + np.arange(9) * 3.0, np.array([-9, -8, -9.5, 1.0, -8.5, -9, -8, -9, -8.2]), np.full(9, 0.5), "r" * 9))
    # 13: one NaN flux
    f = rng.normal(10, 2, 9); f[4] = np.nan
    objs.append((T + np.arange(9) * 3.0, f, np.full(9, 0.5), "r" * 9))
    # 14: unknown filter letters mixed in
    objs.append((T + np.arange(14) * 3.0, rng.normal(10, 2, 14), np.full(14, 0.5), "rrgg??rrggiiii"))
    # 15: clean bright transient, many points, all bands
    t = np.sort(T + rng.uniform(0, 300, 180))
    b = rng.choice(6, 180, p=BAND_PROB)
    f = _bazin(t, 80.0, T + 60, 6.0, 45.0) * (BAND_WAVE_A[2] / BAND_WAVE_A[b]) ** 0.8 + 1.0
    e = BAND_SIGMA[b] * 0.5
    objs.append((t, f + rng.normal(0, 1, 180) * e, e, b))
    # 16: zero flux values (GP flux_scale path) and large point count
    t = np.sort(T + rng.uniform(0, 500, 64))
    f = rng.normal(0, 1, 64); f[::7] = 0.0
    objs.append((t, f, np.full(64, 1.0), rng.choice(6, 64)))
    return from_objects(objs)
