"""Oracle for src/features/colors.py -> 83 columns (order: columns.py ``_color``)."""
import numpy as np

NCOL = 83
PAIRS = [(1, 2), (2, 3), (0, 1), (3, 4)]          # (g,r) (r,i) (u,g) (i,z)   colors.py:31-36
EPOCHS = [0, 10, 20, 30, 50, 75, 100, 150, -10, -20]   # colors.py:154-165


def interpolate_flux(times, fluxes, target_time, max_gap=50.0):
    # colors.py:47-89
    if len(times) < 2:
        return np.nan
    idx = np.argsort(times, kind="stable")
    times, fluxes = times[idx], fluxes[idx]
    if target_time < times[0] or target_time > times[-1]:
        return np.nan
    i = np.searchsorted(times, target_time)
    if i == 0:
        return fluxes[0]
    if i == len(times):
        return fluxes[-1]
    t1, t2 = times[i - 1], times[i]
    f1, f2 = fluxes[i - 1], fluxes[i]
    if t2 - t1 > max_gap:
        return np.nan
    return f1 + (target_time - t1) / (t2 - t1) * (f2 - f1)


def compute_color(f1, f2):
    # colors.py:92-105
    if np.isnan(f1) or np.isnan(f2) or f1 <= 0 or f2 <= 0:
        return np.nan
    return -2.5 * np.log10(f1 / f2)


def _temp(c):
    # colors.py:297-308
    if np.isnan(c):
        return np.nan
    if c < -0.5:
        return 50000.0
    if c > 2.0:
        return 3000.0
    return 7000 / (c + 0.6)


def extract_one(o):
    out = np.full(NCOL, np.nan)
    bd = {k: o.band(k) for k in range(6) if len(o.band(k)[0]) > 0}      # :121-129 file order
    peak_times = {}
    for k in (2, 1, 3):                                                  # :133-138
        if k in bd:
            t, f, _ = bd[k]
            peak_times[k] = t[np.argmax(f)]
    ref = np.nan                                                         # :141-148
    for k in (2, 1, 3):
        if k in peak_times and not np.isnan(peak_times[k]):
            ref = peak_times[k]
            break
    out[0] = ref
    col = {}
    c = 1
    for ei, d in enumerate(EPOCHS):                                      # :168-189
        target = ref + d if not np.isnan(ref) else np.nan
        ef = [interpolate_flux(bd[k][0], bd[k][1], target) if k in bd else np.nan for k in range(6)]
        for pi, (a, b) in enumerate(PAIRS):
            v = compute_color(ef[a], ef[b])
            col[(pi, ei)] = v
            out[c] = v
            c += 1
    for pi in range(4):                                                  # :192-207
        p, c50, c100 = col[(pi, 0)], col[(pi, 4)], col[(pi, 6)]
        out[c] = (c50 - p) / 50.0 if not (np.isnan(p) or np.isnan(c50)) else np.nan
        out[c + 1] = (c100 - p) / 100.0 if not (np.isnan(p) or np.isnan(c100)) else np.nan
        c += 2
    for a, b in PAIRS:                                                   # :211-232
        if a in bd and b in bd:
            cs = []
            for t, f1 in zip(bd[a][0], bd[a][1]):
                v = compute_color(f1, interpolate_flux(bd[b][0], bd[b][1], t, max_gap=5.0))
                if not np.isnan(v):
                    cs.append(v)
            if len(cs) >= 3:
                out[c] = np.std(cs)
                out[c + 1] = np.max(cs) - np.min(cs)
        c += 2
    pk = [np.max(bd[k][1]) if k in bd else np.nan for k in range(6)]     # :235-239
    out[c:c + 6] = pk
    c += 6
    for a, b in PAIRS:                                                   # :242-248
        if not np.isnan(pk[a]) and not np.isnan(pk[b]) and pk[b] > 0:
            out[c] = pk[a] / pk[b]
        c += 1
    for a, b in ((1, 2), (2, 3)):                                        # :252-257
        if a in peak_times and b in peak_times:
            out[c] = peak_times[a] - peak_times[b]
        c += 1
    for pi in (0, 1):                                                    # :263-275
        p, c30, c75 = col[(pi, 0)], col[(pi, 3)], col[(pi, 5)]
        if not any(np.isnan([p, c30, c75])):
            out[c] = ((c75 - c30) / 45.0 - (c30 - p) / 30.0) / 37.5
        c += 1
    for pi in (0, 1):                                                    # :279-293
        late = [col[(pi, 4)], col[(pi, 5)], col[(pi, 6)], col[(pi, 7)]]
        v = [x for x in late if not np.isnan(x)]
        if len(v) >= 2:
            out[c] = np.std(v)
            out[c + 1] = np.mean(v)
        c += 2
    tp, t30, t75, t150 = (_temp(col[(0, e)]) for e in (0, 3, 5, 7))      # :310-313
    out[c:c + 4] = tp, t30, t75, t150
    c += 4
    out[c] = (t30 - tp) / 30.0 if not (np.isnan(tp) or np.isnan(t30)) else np.nan      # :321-334
    out[c + 1] = (t75 - t30) / 45.0 if not (np.isnan(t30) or np.isnan(t75)) else np.nan
    out[c + 2] = (t150 - t75) / 75.0 if not (np.isnan(t75) or np.isnan(t150)) else np.nan
    v = [x for x in (tp, t30, t75, t150) if not np.isnan(x)]             # :337-342
    out[c + 3] = np.std(v) / np.mean(v) if len(v) >= 2 else np.nan
    assert c + 4 == NCOL
    return out
