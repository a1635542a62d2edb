"""Oracle for src/features/tde_physics.py -> 25 columns (order: columns.py ``_tde``)."""
import warnings

import numpy as np

NCOL = 25


def _nearest_pairs(t1, f1, t2, f2, max_dt):
    """Rows of band1 (time order) matched to the nearest band2 row within ``max_dt`` days."""
    for a, fa in zip(t1, f1):
        dt = np.abs(t2 - a)
        j = np.argmin(dt)
        if dt[j] < max_dt:
            yield a, fa, f2[j]


def color_variance(o, k1, k2):
    # tde_physics.py:25-90
    out = np.full(3, np.nan)
    t1, f1, _ = o.band_sorted(k1)
    t2, f2, _ = o.band_sorted(k2)
    if len(t1) < 3 or len(t2) < 3:
        return out
    colors, times = [], []
    for a, fa, fb in _nearest_pairs(t1, f1, t2, f2, 5):
        if fa > 0 and fb > 0:
            colors.append(-2.5 * np.log10(fa / fb))
            times.append(a)
    if len(colors) >= 3:
        colors, times = np.array(colors), np.array(times)
        out[0] = np.var(colors)
        out[1] = np.max(colors) - np.min(colors)
        out[2] = np.polyfit(times - times[0], colors, 1)[0] * 100
    return out


def late_time(o, k):
    # tde_physics.py:93-155
    out = np.full(3, np.nan)
    t, f, _ = o.band_sorted(k)
    if len(t) < 5:
        return out
    pi = np.argmax(f)
    pt, pf = t[pi], f[pi]
    late = t > (pt + 50)
    lt, lf = t[late], f[late]
    if len(lt) >= 3 and pf > 0:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with np.errstate(all="ignore"):
                log_t = np.log10(lt - pt + 1)
                log_f = np.log10(np.maximum(lf, 1e-10))
                if np.std(log_t) > 0:
                    out[0] = np.polyfit(log_t, log_f, 1)[0]
        out[1] = np.mean(lf) / pf
        lmax, lmean = np.max(lf), np.mean(lf)
        out[2] = lmax / lmean if lmean > 0 else 1.0
    return out


def rise_characteristics(o, k):
    # tde_physics.py:158-206
    out = np.full(2, np.nan)
    t, f, _ = o.band_sorted(k)
    if len(t) < 5:
        return out
    pi = np.argmax(f)
    pf = f[pi]
    rt, rf = t[:pi + 1], f[:pi + 1]
    if len(rt) >= 3 and pf > 0:
        norm_flux = rf / pf
        norm_time = (rt - rt[0]) / (rt[-1] - rt[0] + 1e-6)
        out[0] = np.mean(norm_flux) / np.mean(norm_time) if np.mean(norm_time) > 0 else 1.0
        if rt[-1] > rt[0]:
            out[1] = pf / (rt[-1] - rt[0])
    return out


def temperature_stability(o):
    # tde_physics.py:209-284
    out = np.full(3, np.nan)
    tg, fg, _ = o.band_sorted(1)
    tr, fr, _ = o.band_sorted(2)
    if len(tg) < 3 or len(tr) < 3:
        return out
    temps, times = [], []
    for a, g_flux, r_flux in _nearest_pairs(tg, fg, tr, fr, 3):
        if g_flux > 0 and r_flux > 0:
            g_r = -2.5 * np.log10(g_flux / r_flux)
            if g_r < -0.5:
                temp = 40000
            elif g_r > 1.5:
                temp = 5000
            else:
                temp = 7000 / (g_r + 0.5)
            temps.append(temp)
            times.append(a)
    if len(temps) >= 3:
        temps, times = np.array(temps, float), np.array(times)
        out[0] = np.std(temps) / np.mean(temps)
        out[1] = np.polyfit(times - times[0], temps, 1)[0] * 100
        peak_idx = len(temps) // 4
        if len(temps) > 4:
            out[2] = np.mean(temps[-3:]) / np.mean(temps[:max(2, peak_idx)])
    return out


def decay_power_law(o, k=2):
    """(alpha, residual, alpha_late); tde_physics.py:287-352."""
    out = np.full(3, np.nan)
    t, f, _ = o.band_sorted(k)
    if len(t) < 5:
        return out
    pi = np.argmax(f)
    pt, pf = t[pi], f[pi]
    post = t > pt
    ptm, pfl = t[post], f[post]
    if len(ptm) >= 4 and pf > 0:
        dt = np.maximum(ptm - pt, 1)
        valid = pfl > 0
        if np.sum(valid) >= 3:
            log_t = np.log10(dt[valid])
            log_f = np.log10(pfl[valid])
            c = np.polyfit(log_t, log_f, 1)
            out[0] = c[0]
            out[1] = np.std(log_f - (c[0] * log_t + c[1]))
            lv = (dt > 50) & valid
            if np.sum(lv) >= 3:
                out[2] = np.polyfit(np.log10(dt[lv]), np.log10(pfl[lv]), 1)[0]
    return out


def extract_one(o):
    # tde_physics.py:355-374
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return np.concatenate([
            color_variance(o, 1, 2), color_variance(o, 2, 3),
            late_time(o, 1), late_time(o, 2), late_time(o, 3),
            rise_characteristics(o, 1), rise_characteristics(o, 2),
            temperature_stability(o), decay_power_law(o, 2)])
