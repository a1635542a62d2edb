// tde.hpp -- TDE physics features (reference: src/features/tde_physics.py) -> 25 columns.
//
// All inputs are the band-partitioned, time-sorted LDS views of stage.hpp (the reference sorts
// every band with .sort_values('Time (MJD)') first: tde_physics.py:39-40,104,168,221-222,297).
// np.polyfit(x, y, 1) is replaced by the closed-form least-squares line on centred sums; the two
// agree to rounding unless x is (numerically) constant, where the reference itself fails.
#pragma once
#include "stage.hpp"

namespace lcfe {

constexpr int TDE_NCOL = 25;

// Closed-form degree-1 fit of m wave-shared points.  slope/intercept are wave-uniform; Sxx == 0 ->
// NaN (np.polyfit divides by a zero column norm there).
template <class W>
LCFE_FN void wave_linfit(const double* x, const double* y, int m, double& slope, double& icpt) {
    double sx = 0, sy = 0;
    for (int i = W::lane(); i < m; i += W::LANES) { sx += x[i]; sy += y[i]; }
    sx = W::sum(sx);
    sy = W::sum(sy);
    const double mx = sx / m, my = sy / m;
    double sxx = 0, sxy = 0;
    for (int i = W::lane(); i < m; i += W::LANES) {
        const double dx = x[i] - mx;
        sxx += dx * dx;
        sxy += dx * (y[i] - my);
    }
    sxx = W::sum(sxx);
    sxy = W::sum(sxy);
    slope = (sxx > 0) ? sxy / sxx : qnan();
    icpt = my - slope * mx;
}

// Degree-1 fit of (log10(xraw[i]), y[i]) over the elements with xraw[i] > thr.
template <class W>
LCFE_FN void wave_linfit_log10x(const double* xraw, const double* y, int m, double thr, double& slope,
                                double& icpt, int& nsel) {
    double sx = 0, sy = 0;
    int c = 0;
    for (int i = W::lane(); i < m; i += W::LANES)
        if (xraw[i] > thr) { sx += log10(xraw[i]); sy += y[i]; ++c; }
    sx = W::sum(sx);
    sy = W::sum(sy);
    c = W::sum(c);
    nsel = c;
    if (c == 0) { slope = qnan(); icpt = qnan(); return; }
    const double mx = sx / c, my = sy / c;
    double sxx = 0, sxy = 0;
    for (int i = W::lane(); i < m; i += W::LANES)
        if (xraw[i] > thr) {
            const double dx = log10(xraw[i]) - mx;
            sxx += dx * dx;
            sxy += dx * (y[i] - my);
        }
    sxx = W::sum(sxx);
    sxy = W::sum(sxy);
    slope = (sxx > 0) ? sxy / sxx : qnan();
    icpt = my - slope * mx;
}

// mean / population std / min / max of m wave-shared values (np.mean, np.std, np.min, np.max)
template <class W>
LCFE_FN void wave_moments(const double* x, int m, double& mean, double& var, double& mn, double& mx) {
    double s = 0, lo = __builtin_inf(), hi = -__builtin_inf();
    for (int i = W::lane(); i < m; i += W::LANES) {
        const double v = x[i];
        s += v;
        lo = (v < lo) ? v : lo;
        hi = (v > hi) ? v : hi;
    }
    s = W::sum(s);
    mn = W::min(lo);
    mx = W::max(hi);
    mean = s / m;
    double q = 0;
    for (int i = W::lane(); i < m; i += W::LANES) { const double d = x[i] - mean; q += d * d; }
    var = W::sum(q) / m;
}

// index of the element of the time-sorted band (t2, n2) nearest to `a` (np.argmin(|t2 - a|): first
// minimum).  Per-lane serial scan: each lane asks for its own `a`.
LCFE_FN int nearest_index(const double* t2, int n2, double a) {
    int best = 0;
    double bd = fabs(t2[0] - a);
    for (int j = 1; j < n2; ++j) {
        const double d = fabs(t2[j] - a);
        if (d < bd) { bd = d; best = j; }
    }
    return best;
}

// Compact the lanes' (flag, tval, cval) triples, taken in increasing element order, into the
// wave-shared arrays xs/ys; returns the running count.  All lanes call it once per chunk.
template <class W>
LCFE_FN int wave_compact(bool flag, double xv, double yv, double* xs, double* ys, int count) {
    const unsigned long long mask = W::ballot(flag);
    if (flag) {
        const int pos = count + W::prefix(mask);
        xs[pos] = xv;
        ys[pos] = yv;
    }
    return count + popcll(mask);
}

template <int CAP>
struct TdeLds {
    double xs[CAP];
    double ys[CAP];
    double out[TDE_NCOL];
};

// tde_physics.py:25-90 -> (var, range, trend)
template <class W, int CAP>
LCFE_FN void tde_color_variance(const ObjLds<CAP>& L, int k1, int k2, TdeLds<CAP>& S, double* o3) {
    const int lane = W::lane();
    const int s1 = L.boff[k1], n1 = L.boff[k1 + 1] - s1, s2 = L.boff[k2], n2 = L.boff[k2 + 1] - s2;
    if (n1 < 3 || n2 < 3) {                                     // :42-46
        if (lane == 0) { o3[0] = qnan(); o3[1] = qnan(); o3[2] = qnan(); }
        return;
    }
    int cnt = 0;
    for (int base = 0; base < n1; base += W::LANES) {           // :52-66, band1 rows in time order
        const int i = base + lane;
        bool ok = false;
        double col = 0, tt = 0;
        if (i < n1) {
            tt = L.bt[s1 + i];
            const double f1 = L.bf[s1 + i];
            const int j = nearest_index(L.bt + s2, n2, tt);
            if (fabs(L.bt[s2 + j] - tt) < 5) {
                const double f2 = L.bf[s2 + j];
                if (f1 > 0 && f2 > 0) { ok = true; col = -2.5 * log10(f1 / f2); }
            }
        }
        cnt = wave_compact<W>(ok, tt, col, S.xs, S.ys, cnt);
    }
    W::sync();
    if (cnt >= 3) {                                             // :68-84
        double mean, var, mn, mx, slope, icpt;
        wave_moments<W>(S.ys, cnt, mean, var, mn, mx);
        const double t0 = S.xs[0];
        W::sync();
        for (int i = lane; i < cnt; i += W::LANES) S.xs[i] -= t0;
        W::sync();
        wave_linfit<W>(S.xs, S.ys, cnt, slope, icpt);
        if (lane == 0) { o3[0] = var; o3[1] = mx - mn; o3[2] = slope * 100; }
    } else if (lane == 0) {
        o3[0] = qnan(); o3[1] = qnan(); o3[2] = qnan();
    }
    W::sync();
}

// tde_physics.py:93-155 -> (late_slope, late_flux_ratio, rebrightening)
template <class W, int CAP>
LCFE_FN void tde_late_time(const ObjLds<CAP>& L, int k, TdeLds<CAP>& S, double* o3) {
    const int lane = W::lane();
    const int s = L.boff[k], n = L.boff[k + 1] - s;
    if (lane == 0) { o3[0] = qnan(); o3[1] = qnan(); o3[2] = qnan(); }
    if (n < 5) return;                                          // :106-110
    const double* t = L.bt + s;
    const double* f = L.bf + s;
    const int pk = wave_argmax_first<W>(f, n);
    const double pt = t[pk], pf = f[pk];
    int cnt = 0;
    for (int base = 0; base < n; base += W::LANES) {            // :121-123 late = t > peak + 50
        const int i = base + lane;
        const bool ok = (i < n) && (t[i] > pt + 50);
        cnt = wave_compact<W>(ok, ok ? t[i] : 0.0, ok ? f[i] : 0.0, S.xs, S.ys, cnt);
    }
    W::sync();
    if (cnt >= 3 && pf > 0) {                                   // :125
        double mean, var, mn, mx;
        wave_moments<W>(S.ys, cnt, mean, var, mn, mx);          // mean/max of the late fluxes
        W::sync();
        for (int i = lane; i < cnt; i += W::LANES) {            // :131-132
            S.xs[i] = log10(S.xs[i] - pt + 1);
            S.ys[i] = log10(fmax(S.ys[i], 1e-10));
        }
        W::sync();
        double lm, lvar, lmn, lmx, slope, icpt;
        wave_moments<W>(S.xs, cnt, lm, lvar, lmn, lmx);
        wave_linfit<W>(S.xs, S.ys, cnt, slope, icpt);
        if (lane == 0) {
            o3[0] = (sqrt(lvar) > 0) ? slope : qnan();          // :134-138
            o3[1] = mean / pf;                                  // :141
            o3[2] = (mean > 0) ? mx / mean : 1.0;               // :144-147
        }
    }
    W::sync();
}

// tde_physics.py:158-206 -> (rise_shape, rise_rate)
template <class W, int CAP>
LCFE_FN void tde_rise(const ObjLds<CAP>& L, int k, double* o2) {
    const int lane = W::lane();
    const int s = L.boff[k], n = L.boff[k + 1] - s;
    if (lane == 0) { o2[0] = qnan(); o2[1] = qnan(); }
    if (n < 5) return;
    const double* t = L.bt + s;
    const double* f = L.bf + s;
    const int pk = wave_argmax_first<W>(f, n);
    const double pf = f[pk];
    const int nr = pk + 1;                                      // :182-183 rows up to and including the peak
    if (nr >= 3 && pf > 0) {
        const double t0 = t[0], span = t[pk] - t0;
        double sf = 0, st = 0;
        for (int i = lane; i < nr; i += W::LANES) {
            sf += f[i] / pf;                                    // :187
            st += (t[i] - t0) / (span + 1e-6);                  // :188
        }
        sf = W::sum(sf) / nr;
        st = W::sum(st) / nr;
        if (lane == 0) {
            o2[0] = (st > 0) ? sf / st : 1.0;                   // :194
            o2[1] = (t[pk] > t0) ? pf / span : qnan();          // :198-201
        }
    }
    W::sync();
}

// tde_physics.py:209-284 -> (temp_stability, temp_trend, temp_late_vs_peak)
template <class W, int CAP>
LCFE_FN void tde_temperature(const ObjLds<CAP>& L, TdeLds<CAP>& S, double* o3) {
    const int lane = W::lane();
    const int sg = L.boff[1], ng = L.boff[2] - sg, sr = L.boff[2], nr = L.boff[3] - sr;
    if (lane == 0) { o3[0] = qnan(); o3[1] = qnan(); o3[2] = qnan(); }
    if (ng < 3 || nr < 3) return;                               // :224-228
    int cnt = 0;
    for (int base = 0; base < ng; base += W::LANES) {           // :234-258
        const int i = base + lane;
        bool ok = false;
        double temp = 0, tt = 0;
        if (i < ng) {
            tt = L.bt[sg + i];
            const double gf = L.bf[sg + i];
            const int j = nearest_index(L.bt + sr, nr, tt);
            const double rf = L.bf[sr + j];
            if (fabs(L.bt[sr + j] - tt) < 3 && gf > 0 && rf > 0) {
                const double g_r = -2.5 * log10(gf / rf);
                temp = (g_r < -0.5) ? 40000.0 : ((g_r > 1.5) ? 5000.0 : 7000 / (g_r + 0.5));
                ok = true;
            }
        }
        cnt = wave_compact<W>(ok, tt, temp, S.xs, S.ys, cnt);
    }
    W::sync();
    if (cnt >= 3) {                                             // :260-278
        double mean, var, mn, mx, slope, icpt;
        wave_moments<W>(S.ys, cnt, mean, var, mn, mx);
        const double t0 = S.xs[0];
        W::sync();
        for (int i = lane; i < cnt; i += W::LANES) S.xs[i] -= t0;
        W::sync();
        wave_linfit<W>(S.xs, S.ys, cnt, slope, icpt);
        if (lane == 0) {
            o3[0] = sqrt(var) / mean;
            o3[1] = slope * 100;
            if (cnt > 4) {
                const int pi = cnt / 4;
                const int np_ = (pi > 2) ? pi : 2;              // temps[:max(2, peak_idx)]
                double a = 0, b = 0;
                for (int i = 0; i < np_; ++i) a += S.ys[i];
                for (int i = cnt - 3; i < cnt; ++i) b += S.ys[i];
                o3[2] = (b / 3) / (a / np_);
            }
        }
    }
    W::sync();
}

// tde_physics.py:287-352 (band r) -> (alpha, residual, alpha_late)
template <class W, int CAP>
LCFE_FN void tde_decay(const ObjLds<CAP>& L, int k, TdeLds<CAP>& S, double* o3) {
    const int lane = W::lane();
    const int s = L.boff[k], n = L.boff[k + 1] - s;
    if (lane == 0) { o3[0] = qnan(); o3[1] = qnan(); o3[2] = qnan(); }
    if (n < 5) return;
    const double* t = L.bt + s;
    const double* f = L.bf + s;
    const int pk = wave_argmax_first<W>(f, n);
    const double pt = t[pk], pf = f[pk];
    int npost = 0, cnt = 0;
    for (int base = 0; base < n; base += W::LANES) {            // :313-325
        const int i = base + lane;
        const bool post = (i < n) && (t[i] > pt);
        npost += popcll(W::ballot(post));
        const bool ok = post && (f[i] > 0);
        const double dt = ok ? fmax(t[i] - pt, 1.0) : 1.0;
        cnt = wave_compact<W>(ok, dt, ok ? log10(f[i]) : 0.0, S.xs, S.ys, cnt);   // xs = dt (raw), ys = log f
    }
    W::sync();
    if (npost >= 4 && pf > 0 && cnt >= 3) {                     // :317,:323
        // xs holds the raw dt (the late subset is selected on it, :335), x = log10(dt) on the fly
        double slope, icpt, sl = qnan(), ic;
        int nall, nlate;
        wave_linfit_log10x<W>(S.xs, S.ys, cnt, -1.0, slope, icpt, nall);
        // :331-332 np.std(log_f - predicted)
        double sr = 0;
        for (int i = lane; i < cnt; i += W::LANES) sr += S.ys[i] - (slope * log10(S.xs[i]) + icpt);
        const double mr = W::sum(sr) / cnt;
        double q = 0;
        for (int i = lane; i < cnt; i += W::LANES) {
            const double d = (S.ys[i] - (slope * log10(S.xs[i]) + icpt)) - mr;
            q += d * d;
        }
        q = W::sum(q);
        wave_linfit_log10x<W>(S.xs, S.ys, cnt, 50.0, sl, ic, nlate);       // :335-340 (dt > 50) & valid
        if (nlate < 3) sl = qnan();
        if (lane == 0) { o3[0] = slope; o3[1] = sqrt(q / cnt); o3[2] = sl; }
    }
    W::sync();
}

template <class W, int CAP>
LCFE_FN void tde_object(const ObjLds<CAP>& L, TdeLds<CAP>& S) {
    double* o = S.out;                                          // tde_physics.py:355-374
    tde_color_variance<W, CAP>(L, 1, 2, S, o + 0);
    tde_color_variance<W, CAP>(L, 2, 3, S, o + 3);
    tde_late_time<W, CAP>(L, 1, S, o + 6);
    tde_late_time<W, CAP>(L, 2, S, o + 9);
    tde_late_time<W, CAP>(L, 3, S, o + 12);
    tde_rise<W, CAP>(L, 1, o + 15);
    tde_rise<W, CAP>(L, 2, o + 17);
    tde_temperature<W, CAP>(L, S, o + 19);
    tde_decay<W, CAP>(L, 2, S, o + 22);
    W::sync();
}

}  // namespace lcfe
