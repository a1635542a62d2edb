"""Oracle for src/features/bazin_fitting.py.

The bounded fit is ``scipy.optimize.curve_fit`` with the reference's own arguments
(``bazin_fitting.py:128-137``): with bounds scipy runs Trust-Region-Reflective with a 2-point
finite-difference Jacobian.  scipy is the reference's numerical
dependency and is installed here and on the GPU box, so the oracle calls it rather than
re-deriving it; the HIP kernel is the independent restatement of TRF.
"""
import warnings

import numpy as np
from scipy.optimize import OptimizeWarning, curve_fit

NCOL = 52


def bazin_function(t, A, t0, tau_rise, tau_fall, B):
    # bazin_fitting.py:37-60
    numerator = np.exp(-(t - t0) / tau_fall)
    denominator = 1.0 + np.exp(-(t - t0) / tau_rise)
    return A * numerator / denominator + B


def fit_single_band(times, fluxes, flux_errors, info=None):
    """8 values (A, t0, tau_rise, tau_fall, B, chi2, ratio, peak); bazin_fitting.py:63-179."""
    nan8 = np.full(8, np.nan)
    if len(times) < 5:                                         # :76-87
        return nan8
    idx = np.argsort(times, kind="stable")                     # :90-93
    times, fluxes, flux_errors = times[idx], fluxes[idx], flux_errors[idx]
    peak_idx = np.argmax(fluxes)                               # :97-105
    t0_guess = times[peak_idx]
    A_guess = fluxes[peak_idx] - np.median(fluxes)
    B_guess = np.median(fluxes)
    duration = times[-1] - times[0]
    max_flux = np.max(fluxes)
    bounds = ([0, times[0], 0.1, 0.1, -max_flux],              # :114-118
              [3 * max_flux, times[-1], duration, duration, 2 * max_flux])
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", OptimizeWarning)
            warnings.simplefilter("ignore", RuntimeWarning)
            sigma = np.where(flux_errors > 0, flux_errors, 1.0)   # :126
            res = curve_fit(bazin_function, times, fluxes,
                            p0=[A_guess, t0_guess, duration * 0.2, duration * 0.3, B_guess],
                            bounds=bounds, sigma=sigma, absolute_sigma=True, maxfev=2000,
                            full_output=info is not None)
            popt = res[0]
            if info is not None:
                info["nfev"] = res[2]["nfev"]
            A, t0, tau_rise, tau_fall, B = popt
            A = np.clip(A, -1e6, 1e6)                          # :142-145
            tau_rise = np.clip(tau_rise, 0.1, 1e4)
            tau_fall = np.clip(tau_fall, 0.1, 1e4)
            B = np.clip(B, -1e6, 1e6)
            fitted = bazin_function(times, A, t0, tau_rise, tau_fall, B)
            chi2 = np.sum(((fluxes - fitted) / sigma) ** 2)
            reduced = np.clip(chi2 / (len(times) - 5), 0, 1e6)  # :151
            ratio = np.clip(tau_rise / (tau_fall + 1e-6), 0, 100)
            peak = np.clip(A + B, -1e6, 1e6)
            return np.array([A, t0, tau_rise, tau_fall, B, reduced, ratio, peak])
    except (RuntimeError, ValueError, OptimizeWarning):        # :168-179
        return nan8


def extract_one(o):
    """bazin_fitting.py:182-251 -> float64[52]."""
    out = np.full(NCOL, np.nan)
    for k in range(6):
        t, f, e = o.band_sorted(k)                             # :195
        if len(t) < 5:
            continue
        out[8 * k:8 * k + 8] = fit_single_band(t, f, e)
    rise = [out[8 * k + 2] for k in (1, 2, 3) if not np.isnan(out[8 * k + 2])]   # :217-225
    fall = [out[8 * k + 3] for k in (1, 2, 3) if not np.isnan(out[8 * k + 3])]
    if len(rise) >= 2:
        out[48] = np.std(rise) / np.mean(rise)
    if len(fall) >= 2:
        out[49] = np.std(fall) / np.mean(fall)
    chi = [out[8 * k + 5] for k in range(6) if not np.isnan(out[8 * k + 5])]     # :238-249
    if chi:
        out[50] = np.mean(chi)
        out[51] = np.std(chi)
    return out
