"""Oracle for src/features/lightcurve_shape.py -> 65 columns (order: columns.py ``_shape``)."""
import warnings

import numpy as np

NCOL = 65


def rise_time(t, f, pt, pf, frac=0.1):
    # lightcurve_shape.py:34-65
    if np.isnan(pt) or np.isnan(pf) or len(t) < 2:
        return np.nan
    pre = t < pt
    if not np.any(pre):
        return np.nan
    pt_, pf_ = t[pre], f[pre]
    above = pf_ > frac * pf
    if not np.any(above):
        return pt - pt_[0]
    return pt - pt_[np.argmax(above)]


def fade_time(t, f, pt, pf, frac=0.5):
    # lightcurve_shape.py:68-104
    if np.isnan(pt) or np.isnan(pf) or len(t) < 2:
        return np.nan
    post = t > pt
    if not np.any(post):
        return np.nan
    qt, qf = t[post], f[post]
    o = np.argsort(qt, kind="stable")
    qt, qf = qt[o], qf[o]
    below = qf < frac * pf
    if not np.any(below):
        return qt[-1] - pt
    return qt[np.argmax(below)] - pt


def power_law_decay(t, f, pt, pf):
    # lightcurve_shape.py:107-144
    if np.isnan(pt) or np.isnan(pf):
        return np.nan, np.nan
    m = (t > pt + 5) & (f > 0)
    if np.sum(m) < 5:
        return np.nan, np.nan
    dt = np.maximum(t[m] - pt, 1.0)
    log_dt = np.log10(dt)
    log_f = np.log10(np.maximum(f[m], 1e-10))
    try:
        c = np.polyfit(log_dt, log_f, 1)
        return c[0], np.sqrt(np.mean((log_f - (c[0] * log_dt + c[1])) ** 2))
    except Exception:
        return np.nan, np.nan


def duration_above(t, f, frac):
    # lightcurve_shape.py:147-161
    if len(t) < 2:
        return np.nan
    above = f > frac * np.max(f)
    if not np.any(above):
        return 0.0
    return np.max(t[above]) - np.min(t[above])


def extract_one(o):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return _extract_one(o)


def _extract_one(o):
    out = np.full(NCOL, np.nan)
    peak_times = []
    for k in range(6):                                            # :204-247
        t, f, _ = o.band_sorted(k)
        if len(t) < 3:
            continue
        pi = np.argmax(f)
        pt, pf = t[pi], f[pi]
        peak_times.append(pt)
        r = rise_time(t, f, pt, pf)
        f50 = fade_time(t, f, pt, pf, 0.5)
        f25 = fade_time(t, f, pt, pf, 0.25)
        a, res = power_law_decay(t, f, pt, pf)
        asym = r / f50 if (not np.isnan(r) and not np.isnan(f50) and f50 > 0) else np.nan
        out[8 * k:8 * k + 8] = (r, f50, f25, asym, duration_above(t, f, 0.5),
                                duration_above(t, f, 0.25), a, res)
    vp = [x for x in peak_times if not np.isnan(x)]               # :252-258
    if len(vp) >= 2:
        out[48] = np.max(vp) - np.min(vp)
        out[49] = np.std(vp)
    rises = [out[8 * k] for k in (1, 2, 3) if not np.isnan(out[8 * k])]          # :261-284
    fades = [out[8 * k + 1] for k in (1, 2, 3) if not np.isnan(out[8 * k + 1])]
    alphas = [out[8 * k + 6] for k in (1, 2, 3) if not np.isnan(out[8 * k + 6])]
    out[50] = np.mean(rises) if rises else np.nan
    out[51] = np.mean(fades) if fades else np.nan
    out[52] = np.mean(alphas) if alphas else np.nan
    if len(rises) >= 2:
        out[53] = np.std(rises) / (np.mean(rises) + 1e-6)
    if len(fades) >= 2:
        out[54] = np.std(fades) / (np.mean(fades) + 1e-6)
    t, f = o.t, o.f                                               # :287-330 file order
    if len(t) >= 5:
        pi = np.argmax(f)
        pt, pf = t[pi], f[pi]
        r = rise_time(t, f, pt, pf)
        f50 = fade_time(t, f, pt, pf, 0.5)
        out[55], out[56] = r, f50
        if not np.isnan(r) and not np.isnan(f50) and f50 > 0:
            out[57] = r / f50
        out[58], out[59] = power_law_decay(t, f, pt, pf)
        out[60:64] = [np.percentile(f, q) for q in (10, 25, 75, 90)]
        if pf > 0:
            out[64] = pf / (np.sum(f) + 1e-6)
    return out
