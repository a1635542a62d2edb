// shape.hpp -- light-curve shape features (reference: src/features/lightcurve_shape.py) -> 65 columns.
#pragma once
#include "fits.hpp"      // wave_median
#include "stage.hpp"
#include "stat.hpp"      // np_lerp
#include "tde.hpp"

namespace lcfe {

constexpr int SHAPE_NCOL = 65;

template <int CAP>
struct ShapeLds {
    double xs[CAP];
    double ys[CAP];
    unsigned long long keys[CAP];
    double sel[8];
    double pkt[6], pkn[6];       // per-band peak time / "band has >= 3 points"
    double out[SHAPE_NCOL + 1];
};

// lightcurve_shape.py:34-65.  "first" means first in ARRAY order (time order for the band views,
// file order for the all-rows view) -- that is what pre_times[0] / np.argmax(above) select.
template <class W>
LCFE_FN double shape_rise_time(const double* t, const double* f, int n, double pt, double pf) {
    if (is_nan(pt) || is_nan(pf) || n < 2) return qnan();
    const double thr = 0.1 * pf;
    int first_pre = 0x7fffffff, first_above = 0x7fffffff;
    for (int i = W::lane(); i < n; i += W::LANES) {
        if (t[i] < pt) {
            first_pre = (i < first_pre) ? i : first_pre;
            if (f[i] > thr) first_above = (i < first_above) ? i : first_above;
        }
    }
    first_pre = W::min(first_pre);
    first_above = W::min(first_above);
    if (first_pre == 0x7fffffff) return qnan();
    return pt - t[(first_above == 0x7fffffff) ? first_pre : first_above];
}

// lightcurve_shape.py:68-104 (order-free form: earliest post-peak time below the threshold, else
// the latest post-peak time)
template <class W>
LCFE_FN double shape_fade_time(const double* t, const double* f, int n, double pt, double pf, double frac) {
    if (is_nan(pt) || is_nan(pf) || n < 2) return qnan();
    const double thr = frac * pf;
    double tmax = -__builtin_inf(), tbelow = __builtin_inf();
    bool any_post = false;
    for (int i = W::lane(); i < n; i += W::LANES) {
        if (t[i] > pt) {
            any_post = true;
            tmax = (t[i] > tmax) ? t[i] : tmax;
            if (f[i] < thr) tbelow = (t[i] < tbelow) ? t[i] : tbelow;
        }
    }
    if (!W::any(any_post)) return qnan();
    tmax = W::max(tmax);
    tbelow = W::min(tbelow);
    return ((tbelow < __builtin_inf()) ? tbelow : tmax) - pt;
}

// lightcurve_shape.py:147-161
template <class W>
LCFE_FN double shape_duration_above(const double* t, const double* f, int n, double fmax_, double frac) {
    if (n < 2) return qnan();
    const double thr = frac * fmax_;
    double lo = __builtin_inf(), hi = -__builtin_inf();
    bool any = false;
    for (int i = W::lane(); i < n; i += W::LANES) {
        if (f[i] > thr) {
            any = true;
            lo = (t[i] < lo) ? t[i] : lo;
            hi = (t[i] > hi) ? t[i] : hi;
        }
    }
    if (!W::any(any)) return 0.0;
    return W::max(hi) - W::min(lo);
}

// lightcurve_shape.py:107-144 -> (alpha, rms residual)
template <class W>
LCFE_FN void shape_power_law(const double* t, const double* f, int n, double pt, double pf, double* xs, double* ys,
                             double& alpha, double& resid) {
    alpha = qnan();
    resid = qnan();
    if (is_nan(pt) || is_nan(pf)) return;
    int cnt = 0;
    for (int base = 0; base < n; base += W::LANES) {
        const int i = base + W::lane();
        const bool ok = (i < n) && (t[i] > pt + 5) && (f[i] > 0);
        double lx = 0, ly = 0;
        if (ok) { lx = log10(fmax(t[i] - pt, 1.0)); ly = log10(fmax(f[i], 1e-10)); }
        cnt = wave_compact<W>(ok, lx, ly, xs, ys, cnt);
    }
    W::sync();
    if (cnt >= 5) {
        double slope, icpt;
        wave_linfit<W>(xs, ys, cnt, slope, icpt);
        double q = 0;
        for (int i = W::lane(); i < cnt; i += W::LANES) { const double d = ys[i] - (slope * xs[i] + icpt); q += d * d; }
        q = W::sum(q);
        alpha = slope;
        resid = sqrt(q / cnt);
        if (is_nan(slope)) resid = qnan();
    }
    W::sync();
}

// np.percentile(x, q) (linear) of m wave-shared values by rank selection; sel = 2 doubles scratch
template <class W>
LCFE_FN double wave_percentile(const double* x, int m, double q, double* sel, unsigned long long* keys) {
    const double quant = q / 100.0;
    const double vi = m * quant + (1.0 - quant) - 1.0;      // numpy _compute_virtual_index, alpha = beta = 1
    int lo = (int)floor(vi);
    int hi = lo + 1;
    lo = lo < 0 ? 0 : (lo > m - 1 ? m - 1 : lo);
    hi = hi < 0 ? 0 : (hi > m - 1 ? m - 1 : hi);
    const double gamma = vi - floor(vi);
    bool nanf = false;
    for (int i = W::lane(); i < m; i += W::LANES) nanf = nanf || is_nan(x[i]);
    {
        const int ranks[2] = {lo, hi};
        wave_select_ranks<W, 2>(x, m, keys, ranks, sel);
    }
    const double r = np_lerp(sel[0], sel[1], gamma);
    const bool any_nan = W::any(nanf);
    W::sync();
    return any_nan ? qnan() : r;
}

// WG: policy of one per-band pass (8-lane groups on the device: six bands side by side); W: whole wave
template <class W, class WG, int CAP>
LCFE_FN void shape_object(const ObjLds<CAP>& L, ShapeLds<CAP>& S) {
    const int lane = W::lane();
    double* o = S.out;
    for (int k = WG::group_id(); k < 6; k += WG::NGROUPS) {     // lightcurve_shape.py:204-247
        const int s = L.boff[k], n = L.boff[k + 1] - s;
        double v[8];
        for (int j = 0; j < 8; ++j) v[j] = qnan();
        double pkt = qnan();
        if (n >= 3) {
            const double* t = L.bt + s;
            const double* f = L.bf + s;
            const int pk = wave_argmax_first<WG>(f, n);
            const double pt = t[pk], pf = f[pk];
            pkt = pt;
            v[0] = shape_rise_time<WG>(t, f, n, pt, pf);
            v[1] = shape_fade_time<WG>(t, f, n, pt, pf, 0.5);
            v[2] = shape_fade_time<WG>(t, f, n, pt, pf, 0.25);
            v[3] = (!is_nan(v[0]) && !is_nan(v[1]) && v[1] > 0) ? v[0] / v[1] : qnan();
            v[4] = shape_duration_above<WG>(t, f, n, pf, 0.5);  // np.max(fluxes) == pf (NaN propagates)
            v[5] = shape_duration_above<WG>(t, f, n, pf, 0.25);
            shape_power_law<WG>(t, f, n, pt, pf, S.xs + s, S.ys + s, v[6], v[7]);
        }
        if (WG::lane() == 0) {
            for (int j = 0; j < 8; ++j) o[8 * k + j] = v[j];
            S.pkt[k] = pkt;                                      // NaN marks "band has < 3 points"
            S.pkn[k] = (n >= 3) ? 1.0 : 0.0;
        }
        WG::sync();
    }
    W::sync();
    if (lane == 0) {
        // :252-258 peak-time spread over the bands that have >= 3 points
        double vp[6];
        int n = 0;
        for (int k = 0; k < 6; ++k) if (S.pkn[k] != 0.0 && !is_nan(S.pkt[k])) vp[n++] = S.pkt[k];
        if (n >= 2) {
            double lo = vp[0], hi = vp[0], m, sd;
            for (int i = 1; i < n; ++i) { lo = fmin(lo, vp[i]); hi = fmax(hi, vp[i]); }
            mean_std_small(vp, n, m, sd);
            o[48] = hi - lo;
            o[49] = sd;
        } else { o[48] = qnan(); o[49] = qnan(); }
        // :261-284 optical (g, r, i) means and consistencies
        double r_[3], f_[3], a_[3];
        int nr = 0, nf = 0, na = 0;
        for (int k = 1; k <= 3; ++k) {
            if (!is_nan(o[8 * k])) r_[nr++] = o[8 * k];
            if (!is_nan(o[8 * k + 1])) f_[nf++] = o[8 * k + 1];
            if (!is_nan(o[8 * k + 6])) a_[na++] = o[8 * k + 6];
        }
        double m, sd;
        o[50] = qnan(); o[51] = qnan(); o[52] = qnan(); o[53] = qnan(); o[54] = qnan();
        if (nr) { mean_std_small(r_, nr, m, sd); o[50] = m; if (nr >= 2) o[53] = sd / (m + 1e-6); }
        if (nf) { mean_std_small(f_, nf, m, sd); o[51] = m; if (nf >= 2) o[54] = sd / (m + 1e-6); }
        if (na) { mean_std_small(a_, na, m, sd); o[52] = m; }
    }
    // :287-330 all rows, FILE order
    const int n = L.n;
    double v[10];
    for (int j = 0; j < 10; ++j) v[j] = qnan();
    if (n >= 5) {
        const int pk = wave_argmax_first<W>(L.f, n);
        const double pt = L.t[pk], pf = L.f[pk];
        v[0] = shape_rise_time<W>(L.t, L.f, n, pt, pf);
        v[1] = shape_fade_time<W>(L.t, L.f, n, pt, pf, 0.5);
        v[2] = (!is_nan(v[0]) && !is_nan(v[1]) && v[1] > 0) ? v[0] / v[1] : qnan();
        shape_power_law<W>(L.t, L.f, n, pt, pf, S.xs, S.ys, v[3], v[4]);
        v[5] = wave_percentile<W>(L.f, n, 10.0, S.sel, S.keys);
        v[6] = wave_percentile<W>(L.f, n, 25.0, S.sel, S.keys);
        v[7] = wave_percentile<W>(L.f, n, 75.0, S.sel, S.keys);
        v[8] = wave_percentile<W>(L.f, n, 90.0, S.sel, S.keys);
        double sum = 0;
        for (int i = lane; i < n; i += W::LANES) sum += L.f[i];
        sum = W::sum(sum);
        v[9] = (pf > 0) ? pf / (sum + 1e-6) : qnan();
    }
    if (lane == 0) for (int j = 0; j < 10; ++j) o[55 + j] = v[j];
    W::sync();
}

}  // namespace lcfe
