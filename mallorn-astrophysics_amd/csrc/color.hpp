// color.hpp -- colour features (reference: src/features/colors.py) -> 83 columns.
#pragma once
#include "stage.hpp"
#include "tde.hpp"

namespace lcfe {

constexpr int COLOR_NCOL = 83;

// colors.py:47-89 on a time-sorted band (t, f, n).  Serial per lane.
LCFE_FN double interpolate_flux(const double* t, const double* f, int n, double target, double max_gap) {
    if (n < 2) return qnan();
    if (target < t[0] || target > t[n - 1]) return qnan();
    // np.searchsorted(times, target) (side='left'): number of elements < target; NaN sorts last
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (t[mid] < target) lo = mid + 1; else hi = mid;
    }
    if (is_nan(target)) lo = n;
    if (lo == 0) return f[0];
    if (lo == n) return f[n - 1];
    const double t1 = t[lo - 1], t2 = t[lo], f1 = f[lo - 1], f2 = f[lo];
    if (t2 - t1 > max_gap) return qnan();
    const double w = (target - t1) / (t2 - t1);
    return f1 + w * (f2 - f1);
}

// colors.py:92-105
LCFE_FN double compute_color(double f1, double f2) {
    if (is_nan(f1) || is_nan(f2) || f1 <= 0 || f2 <= 0) return qnan();
    return -2.5 * log10(f1 / f2);
}

// colors.py:297-308
LCFE_FN double temp_from_gr(double c) {
    if (is_nan(c)) return qnan();
    if (c < -0.5) return 50000.0;
    if (c > 2.0) return 3000.0;
    return 7000 / (c + 0.6);
}

// position (in the band-sorted view) of the band's first maximum IN FILE ORDER (colors.py:39-44
// np.argmax on the unsorted band rows).  -1 if the band is empty.
template <class W, int CAP>
LCFE_FN int band_argmax_file_order(const ObjLds<CAP>& L, int k) {
    const int s = L.boff[k], n = L.boff[k + 1] - s;
    if (n <= 0) return -1;
    const double* f = L.bf + s;
    bool has_nan = false;
    double best = -__builtin_inf();
    for (int i = W::lane(); i < n; i += W::LANES) {
        const double v = f[i];
        has_nan = has_nan || is_nan(v);
        best = (v > best) ? v : best;
    }
    const bool any_nan = W::any(has_nan);
    best = W::max(best);
    int cand = 0x7fffffff;       // encode (file index << 16 | position)
    for (int i = W::lane(); i < n; i += W::LANES) {
        const double v = f[i];
        const bool hit = any_nan ? is_nan(v) : (v == best);
        if (hit) {
            const int key = ((int)L.bidx[s + i] << 14) | i;      // CAP <= 2048 positions, idx < 2048
            cand = (key < cand) ? key : cand;
        }
    }
    cand = W::min(cand);
    return cand & 0x3fff;
}

template <int CAP>
struct ColorLds {
    double xs[CAP];
    double ys[CAP];
    double ef[64];               // interpolated flux per (epoch, band): ef[6*e + k]
    double out[COLOR_NCOL + 1];
};

LCFE_FN void mean_std_list(const double* v, int n, double& mean, double& sd) {
    double s = 0;
    for (int i = 0; i < n; ++i) s += v[i];
    mean = s / n;
    double q = 0;
    for (int i = 0; i < n; ++i) { const double d = v[i] - mean; q += d * d; }
    sd = sqrt(q / n);
}

template <class W, int CAP>
LCFE_FN void color_object(const ObjLds<CAP>& L, ColorLds<CAP>& S) {
    const int lane = W::lane();
    double* o = S.out;
    const int PA[4] = {1, 2, 0, 3}, PB[4] = {2, 3, 1, 4};          // colors.py:31-36 (g,r)(r,i)(u,g)(i,z)
    const int EP[10] = {0, 10, 20, 30, 50, 75, 100, 150, -10, -20};  // colors.py:154-165
    // peak times of r, g, i in FILE order (colors.py:133-138)
    double pt[6];
    bool has[6];
    double pkf[6];
    for (int k = 0; k < 6; ++k) {
        const int s = L.boff[k], n = L.boff[k + 1] - s;
        has[k] = n > 0;
        pt[k] = qnan();
        pkf[k] = qnan();
        if (n > 0) {
            const int p = band_argmax_file_order<W, CAP>(L, k);
            pt[k] = L.bt[s + p];
            // np.max(fluxes) (colors.py:237): NaN if any NaN, else the maximum == flux at argmax
            pkf[k] = L.bf[s + p];
        }
    }
    double ref = qnan();                                            // :141-148
    if (has[2] && !is_nan(pt[2])) ref = pt[2];
    else if (has[1] && !is_nan(pt[1])) ref = pt[1];
    else if (has[3] && !is_nan(pt[3])) ref = pt[3];
    // 60 interpolations (10 epochs x 6 bands), one per lane
    for (int q = lane; q < 60; q += W::LANES) {
        const int e = q / 6, k = q % 6;
        const int s = L.boff[k], n = L.boff[k + 1] - s;
        const double target = is_nan(ref) ? qnan() : ref + EP[e];
        S.ef[q] = (n > 0) ? interpolate_flux(L.bt + s, L.bf + s, n, target, 50.0) : qnan();
    }
    W::sync();
    if (lane == 0) {
        o[0] = ref;
        for (int e = 0; e < 10; ++e)
            for (int p = 0; p < 4; ++p) o[1 + 4 * e + p] = compute_color(S.ef[6 * e + PA[p]], S.ef[6 * e + PB[p]]);
        for (int p = 0; p < 4; ++p) {                               // :192-207
            const double c0 = o[1 + p], c50 = o[1 + 16 + p], c100 = o[1 + 24 + p];
            o[41 + 2 * p] = (!is_nan(c0) && !is_nan(c50)) ? (c50 - c0) / 50.0 : qnan();
            o[42 + 2 * p] = (!is_nan(c0) && !is_nan(c100)) ? (c100 - c0) / 100.0 : qnan();
        }
    }
    // per-observation colours (colors.py:211-232)
    for (int p = 0; p < 4; ++p) {
        const int a = PA[p], b = PB[p];
        const int sa = L.boff[a], na = L.boff[a + 1] - sa, sb = L.boff[b], nb = L.boff[b + 1] - sb;
        int cnt = 0;
        if (na > 0 && nb > 0) {
            for (int base = 0; base < na; base += W::LANES) {
                const int i = base + lane;
                double c = qnan();
                if (i < na) c = compute_color(L.bf[sa + i], interpolate_flux(L.bt + sb, L.bf + sb, nb, L.bt[sa + i], 5.0));
                cnt = wave_compact<W>(!is_nan(c), c, c, S.xs, S.ys, cnt);
            }
        }
        W::sync();
        double sd = qnan(), rg = qnan();
        if (cnt >= 3) {
            double mean, var, mn, mx;
            wave_moments<W>(S.xs, cnt, mean, var, mn, mx);
            sd = sqrt(var);
            rg = mx - mn;
        }
        if (lane == 0) { o[49 + 2 * p] = sd; o[50 + 2 * p] = rg; }
        W::sync();
    }
    if (lane == 0) {
        for (int k = 0; k < 6; ++k) o[57 + k] = pkf[k];              // :235-239
        for (int p = 0; p < 4; ++p) {                               // :242-248
            const double f1 = pkf[PA[p]], f2 = pkf[PB[p]];
            o[63 + p] = (!is_nan(f1) && !is_nan(f2) && f2 > 0) ? f1 / f2 : qnan();
        }
        o[67] = (has[1] && has[2]) ? pt[1] - pt[2] : qnan();        // :252-257
        o[68] = (has[2] && has[3]) ? pt[2] - pt[3] : qnan();
        for (int p = 0; p < 2; ++p) {                               // :263-275
            const double c0 = o[1 + p], c30 = o[1 + 12 + p], c75 = o[1 + 20 + p];
            if (!is_nan(c0) && !is_nan(c30) && !is_nan(c75)) {
                const double s1 = (c30 - c0) / 30.0, s2 = (c75 - c30) / 45.0;
                o[69 + p] = (s2 - s1) / 37.5;
            } else o[69 + p] = qnan();
        }
        for (int p = 0; p < 2; ++p) {                               // :279-293
            const double late[4] = {o[1 + 16 + p], o[1 + 20 + p], o[1 + 24 + p], o[1 + 28 + p]};
            double v[4];
            int n = 0;
            for (int i = 0; i < 4; ++i) if (!is_nan(late[i])) v[n++] = late[i];
            if (n >= 2) { double m, sd; mean_std_list(v, n, m, sd); o[71 + 2 * p] = sd; o[72 + 2 * p] = m; }
            else { o[71 + 2 * p] = qnan(); o[72 + 2 * p] = qnan(); }
        }
        const double tp = temp_from_gr(o[1]), t30 = temp_from_gr(o[1 + 12]), t75 = temp_from_gr(o[1 + 20]),
                     t150 = temp_from_gr(o[1 + 28]);                 // :310-313
        o[75] = tp; o[76] = t30; o[77] = t75; o[78] = t150;
        o[79] = (!is_nan(tp) && !is_nan(t30)) ? (t30 - tp) / 30.0 : qnan();       // :321-334
        o[80] = (!is_nan(t30) && !is_nan(t75)) ? (t75 - t30) / 45.0 : qnan();
        o[81] = (!is_nan(t75) && !is_nan(t150)) ? (t150 - t75) / 75.0 : qnan();
        const double tt[4] = {tp, t30, t75, t150};                  // :337-342
        double v[4];
        int n = 0;
        for (int i = 0; i < 4; ++i) if (!is_nan(tt[i])) v[n++] = tt[i];
        if (n >= 2) { double m, sd; mean_std_list(v, n, m, sd); o[82] = sd / m; } else o[82] = qnan();
    }
    W::sync();
}

}  // namespace lcfe
