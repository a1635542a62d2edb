"""Oracle for src/features/statistical.py (reference lines cited per function)."""
import numpy as np

from .common import BANDS

NCOL = 123


def skewness(x):
    # statistical.py:14-23
    n = len(x)
    if n < 3:
        return 0.0
    mean = np.mean(x)
    std = np.std(x)
    if std == 0:
        return 0.0
    return np.mean(((x - mean) / std) ** 3)


def kurtosis(x):
    # statistical.py:26-35
    n = len(x)
    if n < 4:
        return 0.0
    mean = np.mean(x)
    std = np.std(x)
    if std == 0:
        return 0.0
    return np.mean(((x - mean) / std) ** 4) - 3


def band_statistics(flux, err, t):
    """17 values in column order; statistical.py:41-132."""
    n = len(flux)
    out = np.full(17, np.nan)
    out[0] = n
    if n == 0:                                  # :56-66
        return out
    mean = np.mean(flux)
    std = np.std(flux) if n > 1 else 0.0        # :71
    mn, mx = np.min(flux), np.max(flux)
    med = np.median(flux)
    out[1:6] = mean, std, mn, mx, med
    if n > 2:                                   # :77-82
        out[6] = skewness(flux)
        out[7] = kurtosis(flux)
    else:
        out[6] = out[7] = 0.0
    out[8] = mx - mn
    out[9] = np.median(np.abs(flux - med))      # :86
    out[10] = np.percentile(flux, 75) - np.percentile(flux, 25) if n > 1 else 0.0   # :87
    if std > 0:                                 # :90-96
        zs = np.abs(flux - mean) / std
        out[11] = np.mean(zs > 1)
        out[12] = np.mean(zs > 2)
    else:
        out[11] = out[12] = 0.0
    if n > 1:                                   # :99-113
        idx = np.argsort(t, kind="stable")
        sf, st = flux[idx], t[idx]
        dt, df = np.diff(st), np.diff(sf)
        valid = dt > 0
        out[13] = np.max(np.abs(df[valid] / dt[valid])) if np.any(valid) else 0.0
    else:
        out[13] = 0.0
    ve = err > 0                                # :116-120
    out[14] = np.mean(np.abs(flux[ve]) / err[ve]) if np.any(ve) else np.nan
    if n > 1:                                   # :123-130
        out[15] = np.max(t) - np.min(t)
        out[16] = np.mean(np.diff(np.sort(t)))
    else:
        out[15] = out[16] = 0.0
    return out


def extract_one(o):
    """statistical.py:159-224 for one object -> float64[123]."""
    out = np.empty(NCOL)
    for k in range(6):
        t, f, e = o.band(k)
        out[17 * k:17 * k + 17] = band_statistics(f, e, t)
    out[102:119] = band_statistics(o.f, o.e, o.t)          # :186-192 all rows, file order
    means = out[1:102:17]
    maxes = out[4:102:17]
    with np.errstate(all="ignore"):
        for j, (a, b) in enumerate(((1, 2), (2, 3), (3, 4))):   # :201-214
            out[119 + j] = means[a] / means[b] if (not np.isnan(means[a]) and means[b] > 0) else np.nan
    valid = ~np.isnan(maxes)                                 # :217-222  (first max wins)
    if valid.any():
        best, bi = -np.inf, -1
        for k in range(6):
            if valid[k] and (bi < 0 or maxes[k] > best):
                best, bi = maxes[k], k
        out[122] = bi
    else:
        out[122] = -1
    return out
