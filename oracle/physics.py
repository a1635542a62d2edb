"""Oracle for src/features/physics_based.py -> 32 columns (order: columns.py ``_physics``)."""
import warnings

import numpy as np

NCOL = 32


def stetson_j(t1, f1, e1, t2, f2, e2, max_dt=0.5):
    # physics_based.py:31-82
    if len(t1) < 3 or len(t2) < 3:
        return np.nan
    mean1, std1 = np.mean(f1), np.std(f1)
    mean2, std2 = np.mean(f2), np.std(f2)
    if std1 == 0 or std2 == 0:
        return 0.0
    j_sum, n_pairs = 0.0, 0
    for a, fa, ea in zip(t1, f1, e1):
        dt = np.abs(t2 - a)
        j = np.argmin(dt)
        if dt[j] <= max_dt:
            fb, eb = f2[j], e2[j]
            if ea > 0 and eb > 0:
                d1 = (fa - mean1) / ea
                d2 = (fb - mean2) / eb
                j_sum += np.sign(d1 * d2) * np.sqrt(np.abs(d1 * d2))
                n_pairs += 1
    if n_pairs == 0:
        return np.nan
    return j_sum / n_pairs


def stetson_k(f, e):
    # physics_based.py:85-107
    if len(f) < 4:
        return np.nan
    mean = np.mean(f)
    n = len(f)
    valid = e > 0
    if np.sum(valid) < 4:
        return np.nan
    d = np.abs(f[valid] - mean) / e[valid]
    return np.sum(d) / np.sqrt(np.sum(d ** 2)) / np.sqrt(n)


TAUS = [1, 5, 10, 30, 100]


def structure_function(t, f):
    """5 SF values + slope; physics_based.py:110-168."""
    out = np.full(6, np.nan)
    if len(t) < 5:
        return out
    i, j = np.triu_indices(len(t), k=1)
    dt = np.abs(t[j] - t[i])
    df = (f[j] - f[i]) ** 2
    for q, tau in enumerate(TAUS):
        v = df[(dt >= tau * 0.5) & (dt <= tau * 1.5)]
        if len(v) >= 3:
            out[q] = np.sqrt(np.mean(v))
    lx = [np.log10(tau) for q, tau in enumerate(TAUS) if not np.isnan(out[q]) and out[q] > 0]
    ly = [np.log10(out[q]) for q, tau in enumerate(TAUS) if not np.isnan(out[q]) and out[q] > 0]
    if len(lx) >= 3:
        out[5] = np.polyfit(lx, ly, 1)[0]
    return out


def estimate_temperature(g, r, i):
    # physics_based.py:171-199
    if any(x <= 0 or np.isnan(x) for x in (g, r, i)):
        return np.nan
    c = -2.5 * np.log10(g / r)
    if c < -0.5:
        temp = 50000
    elif c > 2.0:
        temp = 3000
    else:
        temp = 7000 / (c + 0.6)
    return float(np.clip(temp, 3000, 100000))


def bazin_simple(t, f):
    """(amplitude, t0, rise_approx, fall_approx, plateau); physics_based.py:202-289."""
    out = np.full(5, np.nan)
    if len(t) < 5:
        return out
    o = np.argsort(t, kind="stable")
    t, f = t[o], f[o]
    pi = np.argmax(f)
    pt, pf = t[pi], f[pi]
    out[0], out[1] = pf, pt
    if pi + 1 >= 2:                                       # :235-252
        th10, th90 = 0.1 * pf, 0.9 * pf
        t10, t90 = t[0], pt
        for a, fa in zip(t[:pi + 1], f[:pi + 1]):
            if fa >= th10 and t10 == t[0]:
                t10 = a
            if fa >= th90:
                t90 = a
                break
        out[2] = t90 - t10
    qt, qf = t[pi:], f[pi:]                               # :255-274
    if len(qt) >= 3:
        target = pf / np.e
        fall = np.nan
        for a, fa in zip(qt, qf):
            if fa <= target:
                fall = a - pt
                break
        if np.isnan(fall) and len(qt) > 1:
            fall = (qt[-1] - pt) * pf / (pf - qf[-1] + 1e-6)
        out[3] = fall
    if len(qf) >= 5:                                      # :277-287
        mid = len(qf) // 2
        early, late = np.mean(qf[:mid]), np.mean(qf[mid:])
        if early > 0:
            out[4] = late / early
    return out


def extract_one(o):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with np.errstate(all="ignore"):
            return _extract_one(o)


def _extract_one(o):
    out = np.full(NCOL, np.nan)
    bd = {k: o.band_sorted(k) for k in range(6) if len(o.band(k)[0]) >= 3}     # :306-314
    for q, (a, b) in enumerate(((1, 2), (2, 3), (1, 3))):                        # :318-326
        if a in bd and b in bd:
            out[q] = stetson_j(*bd[a], *bd[b])
    for q, k in enumerate((1, 2, 3)):                                            # :329-334
        if k in bd:
            out[3 + q] = stetson_k(bd[k][1], bd[k][2])
    if 2 in bd:                                                                  # :338-345
        out[6:12] = structure_function(bd[2][0], bd[2][1])
    z = o.z if not np.isnan(o.z) else 0                                          # :348
    for q, k in enumerate((1, 2, 3)):                                            # :351-379
        if k in bd:
            t, f, _ = bd[k]
            out[12 + 3 * q] = (t[-1] - t[0]) / (1 + z)
            pi = np.argmax(f)
            if pi > 0:
                out[13 + 3 * q] = (t[pi] - t[0]) / (1 + z)
            if pi < len(t) - 1:
                out[14 + 3 * q] = (t[-1] - t[pi]) / (1 + z)
    if 1 in bd and 2 in bd and 3 in bd:                                          # :383-423
        out[21] = estimate_temperature(np.max(bd[1][1]), np.max(bd[2][1]), np.max(bd[3][1]))
        rt = bd[2][0]
        target = rt[np.argmax(bd[2][1])] + 50
        late = []
        for k in (1, 2, 3):
            dt = np.abs(bd[k][0] - target)
            j = np.argmin(dt)
            late.append(bd[k][1][j] if dt[j] < 20 else np.nan)
        out[22] = estimate_temperature(*late)
        if not np.isnan(out[21]) and not np.isnan(out[22]):
            out[23] = (out[22] - out[21]) / 50.0
    if 2 in bd:                                                                  # :427-433
        out[24:29] = bazin_simple(bd[2][0], bd[2][1])
    f, e = o.f, o.e                                                              # :437-456
    valid = (e > 0) & (f > 0)
    if np.sum(valid) > 0:
        snr = f[valid] / e[valid]
        out[29] = np.mean(snr)
        out[30] = np.median(snr)
        mean_flux = np.mean(f[valid])
        excess = (np.var(f[valid]) - np.mean(e[valid] ** 2)) / mean_flux ** 2
        out[31] = max(0, excess)
    return out
