"""Oracle for src/features/research_features.py (the v115 "research" features) -> 40 columns.

Restated per function of the reference module; np.polyfit, np.percentile, np.interp and scipy.signal.convolve
are kept as the numpy/scipy calls the reference makes, so the oracle has their arithmetic.  Pinned bit-exact
against the real module by tests/test_oracle_golden.py (golden_research.npz, made by
tests/golden/make_research_golden.py).
"""
import numpy as np
from scipy.signal import convolve

NCOL = 40
_PL = ["powerlaw_alpha", "powerlaw_alpha_deviation_53", "powerlaw_alpha_deviation_512", "powerlaw_chi2",
       "powerlaw_residual_std", "powerlaw_fit_success"]
COLUMNS = ([f"{b}_{k}" for b in "gri" for k in _PL]
           + ["optical_mean_powerlaw_alpha", "optical_std_powerlaw_alpha", "optical_mean_deviation_53",
              "nuclear_smoothness", "nuclear_concentration", "nuclear_variability_ratio", "nuclear_position_score",
              "g_r_color_at_peak", "g_r_color_peak_to_late", "r_i_color_at_peak", "r_i_color_peak_to_late",
              "mhps_10d", "mhps_30d", "mhps_100d", "mhps_10_100_ratio", "mhps_30_100_ratio", "mhps_dominant_scale",
              "luminosity_distance_mpc", "peak_luminosity", "luminosity_amplitude", "mean_luminosity",
              "luminosity_decline_rate"])
assert len(COLUMNS) == NCOL

H0, C_KMS, OMEGA_M, OMEGA_L = 70.0, 299792.458, 0.3, 0.7          # research_features.py:27-31


def fit_power_law_decay(t, f, e):
    """research_features.py:44-117 on a time-sorted band with >= 5 rows -> 6 values."""
    out = [np.nan, np.nan, np.nan, np.nan, np.nan, 0.0]
    peak_idx = np.argmax(f)                                              # :70
    peak_time = t[peak_idx]
    post = (t > peak_time + 10) & (f > 0)                                # :75
    pt, pf, pe = t[post], f[post], e[post]
    if len(pt) < 4:                                                      # :80
        return out
    dt = pt - peak_time
    log_t, log_f = np.log10(dt), np.log10(pf)                            # :84-86
    try:
        coeffs, _ = np.polyfit(log_t, log_f, 1, cov=True)                # :90
        alpha = coeffs[0]
        out[0] = alpha
        out[1] = np.abs(alpha - (-5 / 3))
        out[2] = np.abs(alpha - (-5 / 12))
        resid = log_f - (coeffs[0] * log_t + coeffs[1])                  # :98-99
        out[4] = np.std(resid)
        if len(pe) > 2:                                                  # :103-109
            log_errs = np.clip(pe / (pf * np.log(10) + 1e-10), 0.01, 1.0)
            out[3] = np.sum((resid / log_errs) ** 2) / max(len(resid) - 2, 1)
        out[5] = 1.0
    except Exception:
        pass
    return out


def power_law_features(o):
    """research_features.py:120-160 -> 21 values."""
    out = []
    alphas = []
    for k in (1, 2, 3):
        t, f, e = o.band_sorted(k)
        if len(t) < 5:                                                   # :128-133
            out += [np.nan] * 6
            continue
        v = fit_power_law_decay(t, f, e)
        out += v
        if not np.isnan(v[0]):
            alphas.append(v[0])
    if len(alphas) >= 2:                                                 # :147-158
        out += [np.mean(alphas), np.std(alphas), np.mean([np.abs(a - (-5 / 3)) for a in alphas])]
    else:
        out += [alphas[0] if alphas else np.nan, np.nan, np.abs(alphas[0] - (-5 / 3)) if alphas else np.nan]
    return out


def nuclear_features(o):
    """research_features.py:163-247 -> 4 values."""
    sm = conc = ratio = score = np.nan
    t, f, e = o.band_sorted(2)
    if len(t) < 10:                                                      # :185
        return [sm, conc, ratio, score]
    rate = np.abs(np.diff(f)) / (np.diff(t) + 0.1)                       # :193-195
    median_err = np.median(e)
    if median_err > 0:                                                   # :199-201
        sm = 1.0 / (1.0 + np.median(rate) / median_err)
    peak = np.max(f)
    base = np.percentile(f, 10)                                          # :206
    if base > 0:
        conc = peak / base
    elif peak > 0:
        conc = peak / np.median(np.abs(f) + 1)                           # :210-211
    if len(t) >= 20:                                                     # :215-227
        short = [np.std(f[i:i + 5]) for i in range(len(t) - 5) if t[i + 5] - t[i] < 15]
        long_var = np.std(f)
        if len(short) > 0 and long_var > 0:
            ratio = np.mean(short) / long_var
    scores = []                                                          # :230-243
    if not np.isnan(sm):
        scores.append(sm)
    if not np.isnan(conc):
        scores.append(min(1.0, conc / 100))
    if not np.isnan(ratio):
        scores.append(1.0 - min(1.0, ratio))
    if scores:
        score = np.mean(scores)
    return [sm, conc, ratio, score]


def _first_nanargmax(x):
    """pandas Series.idxmax: first maximum, NaN skipped (all-NaN is outside the contract: the reference raises)."""
    return int(np.nanargmax(x))


def color_at_peak(o):
    """research_features.py:250-331 -> 4 values.  Band frames are in FILE order here (no sort in the reference)."""
    out = [np.nan] * 4
    tr, fr, _ = o.band(2)
    if len(tr) < 3:                                                      # :273-279
        tg, fg, _ = o.band(1)
        if len(tg) < 3:
            return out
        peak_time = tg[_first_nanargmax(fg)]
    else:
        peak_time = tr[_first_nanargmax(fr)]
    for p, (k1, k2) in enumerate(((1, 2), (2, 3))):
        t1, f1, _ = o.band(k1)
        t2, f2, _ = o.band(k2)
        if len(t1) < 2 or len(t2) < 2:                                   # :285
            continue
        n1 = np.abs(t1 - peak_time) < 10                                 # :291-292
        n2 = np.abs(t2 - peak_time) < 10
        if n1.sum() == 0 or n2.sum() == 0:
            continue
        a = f1[n1][np.argmin(np.abs(t1[n1] - peak_time))]                # :296-297
        b = f2[n2][np.argmin(np.abs(t2[n2] - peak_time))]
        if not (a > 0 and b > 0):
            continue
        cpk = -2.5 * np.log10(a / b)                                     # :304
        out[2 * p] = cpk
        l1 = t1 > peak_time + 50                                         # :308-309
        l2 = t2 > peak_time + 50
        if l1.sum() > 0 and l2.sum() > 0:
            t2l, f2l = t2[l2], f2[l2]
            cols = []
            for ta, fa in zip(t1[l1], f1[l1]):                           # :314-325
                d = np.abs(t2l - ta)
                j = np.argmin(d)
                if d[j] < 5:
                    fb = f2l[j]
                    if fa > 0 and fb > 0:
                        cols.append(-2.5 * np.log10(fa / fb))
            if cols:
                out[2 * p + 1] = np.mean(cols) - cpk                     # :329
    return out


def mexican_hat_wavelet(scale, length):
    """research_features.py:338-349."""
    t = np.linspace(-length // 2, length // 2, length)
    x = t / scale
    w = (1 - x ** 2) * np.exp(-x ** 2 / 2)
    return w / np.sqrt(np.sum(w ** 2))


def mhps_features(o):
    """research_features.py:352-430 -> 6 values."""
    vals = {10: np.nan, 30: np.nan, 100: np.nan}
    out_tail = [np.nan, np.nan, np.nan]
    t, f, _ = o.band_sorted(2)
    if len(t) < 20 or t[-1] - t[0] < 50:                                 # :379-388
        return [np.nan] * 6
    t_reg = np.arange(t[0], t[-1], 1.0)                                  # :391
    f_reg = np.interp(t_reg, t, f)
    f_reg = f_reg - np.mean(f_reg)                                       # :397
    got = {}
    for scale in (10, 30, 100):
        wl = int(min(5 * scale, len(f_reg) // 2))                        # :403
        if wl < 5:
            continue
        conv = convolve(f_reg, mexican_hat_wavelet(scale, wl), mode="same")   # :410
        p = np.sum(conv ** 2) / len(conv)
        vals[scale] = p
        got[scale] = p
    if 10 in got and 100 in got and got[100] > 0:                        # :419-423
        out_tail[0] = got[10] / got[100]
    if 30 in got and 100 in got and got[100] > 0:
        out_tail[1] = got[30] / got[100]
    if got:
        out_tail[2] = float(max(got, key=got.get))                       # :426-428
    return [vals[10], vals[30], vals[100]] + out_tail


def luminosity_distance_mpc(z):
    """research_features.py:437-458."""
    if z <= 0 or np.isnan(z):
        return np.nan
    if z < 0.1:
        return (C_KMS / H0) * z * (1 + z / 2)
    q0 = 0.5 * OMEGA_M - OMEGA_L
    return (C_KMS / H0) * z * (1 + 0.5 * (1 - q0) * z)


def luminosity_features(o):
    """research_features.py:461-530 (and the Z > 0 guard of :552-559) -> 5 values."""
    out = [np.nan] * 5
    z = o.z
    if not (z > 0):                                                      # :553
        return out
    d_l = luminosity_distance_mpc(z)
    if np.isnan(d_l):
        return out
    out[0] = d_l
    m = (o.b >= 1) & (o.b <= 3)                                          # :488
    if m.sum() < 5:
        return out
    t, f = o.t[m], o.f[m]
    order = np.argsort(t, kind="stable")                                 # :494 (stable: SURVEY.md 8a sort rule)
    t, f = t[order], f[order]
    lum = f * (d_l ** 2)                                                 # :500
    out[1] = np.max(lum)
    out[2] = np.max(lum) - np.percentile(lum, 10)                        # :506-507
    out[3] = np.mean(lum)
    pk = np.argmax(lum)                                                  # :513
    if pk < len(lum) - 5:
        pl, pt = lum[pk:], t[pk:]
        if len(pl) >= 3 and np.min(pl) > 0:                              # :519
            dt = pt - pt[0]
            if np.std(dt) > 0:
                out[4] = np.polyfit(dt, np.log10(pl), 1)[0] * 100        # :525-526
    return out


def extract_one(o):
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return np.array(power_law_features(o) + nuclear_features(o) + color_at_peak(o) + mhps_features(o)
                            + luminosity_features(o), dtype=np.float64)
