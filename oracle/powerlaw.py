"""Oracle for the v55 post-peak decline fits (scripts/train_v55_powerlaw.py:108-202,
identical to scripts/visualize_and_powerlaw.py:148-255).  As for Bazin, the bounded fits are
``scipy.optimize.curve_fit`` calls with the reference's arguments."""
import warnings

import numpy as np
from scipy.optimize import curve_fit

NCOL = 27


def _pl(p):
    def f(t, A, t0):                                           # train_v55_powerlaw.py:108-127
        return A * np.power(np.maximum(t - t0, 0.1), p)
    return f


def exponential(t, A, tau, t0):                                # :129-130
    return A * np.exp(-np.maximum(t - t0, 0) / tau)


def linear(t, A, b, t0):                                       # :132-133
    return A - b * np.maximum(t - t0, 0)


MODELS = [("powerlaw_5_3", _pl(-5 / 3), 2), ("powerlaw_1", _pl(-1), 2), ("powerlaw_1_5", _pl(-1.5), 2),
          ("powerlaw_2", _pl(-2), 2), ("powerlaw_2_5", _pl(-2.5), 2), ("powerlaw_3", _pl(-3), 2),
          ("powerlaw_0_5", _pl(-0.5), 2), ("exponential", exponential, 3), ("linear", linear, 3)]


def fit_decline_models(t, flux, info=None):
    """9 R^2 values for one band's time-sorted (t, flux); train_v55_powerlaw.py:147-194."""
    out = np.full(9, np.nan)
    if len(t) < 5:                                             # :150-151
        return out
    peak_idx = np.argmax(flux)                                 # :157-159
    peak_time, peak_flux = t[peak_idx], flux[peak_idx]
    post = t > peak_time                                       # :161-163
    if np.sum(post) < 3:
        return out
    t_post = t[post] - peak_time
    flux_post = flux[post]
    for j, (name, func, npar) in enumerate(MODELS):
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                if npar == 2:                                  # :172-184
                    p0, bounds = [peak_flux, 0], ([0, -10], [1e6, 10])
                elif name == "exponential":
                    p0, bounds = [peak_flux, 30, 0], ([0, 1, -10], [1e6, 500, 10])
                else:
                    p0, bounds = [peak_flux, 1, 0], ([0, 0, -10], [1e6, 100, 10])
                res = curve_fit(func, t_post, flux_post, p0=p0, maxfev=1000, bounds=bounds,
                                full_output=info is not None)
                popt = res[0]
                if info is not None:
                    info[name] = (popt, res[2]["nfev"])
                pred = func(t_post, *popt)                     # :186-190
                ss_res = np.sum((flux_post - pred) ** 2)
                ss_tot = np.sum((flux_post - np.mean(flux_post)) ** 2)
                out[j] = 1 - (ss_res / ss_tot) if ss_tot > 0 else 0
        except Exception:
            out[j] = np.nan
    return out


def extract_one(o):
    out = np.full(NCOL, np.nan)
    for j, k in enumerate((1, 2, 3)):                          # :198 bands g, r, i
        t, f, _ = o.band_sorted(k)                             # :153 sort_values
        out[9 * j:9 * j + 9] = fit_decline_models(t, f)
    return out
