// physics.hpp -- physics-based features (reference: src/features/physics_based.py) -> 32 columns.
#pragma once
#include "fits.hpp"
#include "stage.hpp"
#include "tde.hpp"

namespace lcfe {

constexpr int PHYSICS_NCOL = 32;

template <int CAP>
struct PhysicsLds {
    double xs[CAP];
    double ys[CAP];
    unsigned long long keys[CAP];
    double slot[2];
    double out[PHYSICS_NCOL];
};

// physics_based.py:171-199
LCFE_FN double estimate_temperature(double g, double r, double i) {
    if (g <= 0 || is_nan(g) || r <= 0 || is_nan(r) || i <= 0 || is_nan(i)) return qnan();
    const double c = -2.5 * log10(g / r);
    const double temp = (c < -0.5) ? 50000.0 : ((c > 2.0) ? 3000.0 : 7000 / (c + 0.6));
    return np_clip(temp, 3000.0, 100000.0);
}

// physics_based.py:31-82 on two time-sorted bands
template <class W, int CAP>
LCFE_FN double stetson_j(const ObjLds<CAP>& L, int k1, int k2) {
    const int s1 = L.boff[k1], n1 = L.boff[k1 + 1] - s1, s2 = L.boff[k2], n2 = L.boff[k2 + 1] - s2;
    if (n1 < 3 || n2 < 3) return qnan();
    double m1, v1, lo, hi, m2, v2;
    wave_moments<W>(L.bf + s1, n1, m1, v1, lo, hi);
    wave_moments<W>(L.bf + s2, n2, m2, v2, lo, hi);
    if (sqrt(v1) == 0 || sqrt(v2) == 0) return 0.0;
    double js = 0;
    int np_ = 0;
    for (int i = W::lane(); i < n1; i += W::LANES) {
        const double a = L.bt[s1 + i], e1 = L.be[s1 + i];
        const int j = nearest_index(L.bt + s2, n2, a);
        if (fabs(L.bt[s2 + j] - a) <= 0.5) {
            const double e2 = L.be[s2 + j];
            if (e1 > 0 && e2 > 0) {
                const double d1 = (L.bf[s1 + i] - m1) / e1, d2 = (L.bf[s2 + j] - m2) / e2;
                const double pr = d1 * d2;
                const double sg = is_nan(pr) ? qnan() : ((pr > 0) ? 1.0 : ((pr < 0) ? -1.0 : 0.0));
                js += sg * sqrt(fabs(pr));
                ++np_;
            }
        }
    }
    js = W::sum(js);
    np_ = W::sum(np_);
    return (np_ == 0) ? qnan() : js / np_;
}

// physics_based.py:85-107
template <class W>
LCFE_FN double stetson_k(const double* f, const double* e, int n) {
    if (n < 4) return qnan();
    double s = 0;
    for (int i = W::lane(); i < n; i += W::LANES) s += f[i];
    const double mean = W::sum(s) / n;
    double sd = 0, sd2 = 0;
    int nv = 0;
    for (int i = W::lane(); i < n; i += W::LANES) {
        if (e[i] > 0) {
            const double d = fabs(f[i] - mean) / e[i];
            sd += d;
            sd2 += d * d;
            ++nv;
        }
    }
    nv = W::sum(nv);
    if (nv < 4) return qnan();
    return W::sum(sd) / sqrt(W::sum(sd2)) / sqrt((double)n);
}

// physics_based.py:110-168 -> 5 SF values + log-log slope (all N(N-1)/2 pairs of the band)
template <class W>
LCFE_FN void structure_function(const double* t, const double* f, int n, double* out6) {
    const double TAU[5] = {1, 5, 10, 30, 100};
    double sf[5];
    for (int q = 0; q < 5; ++q) sf[q] = qnan();
    if (n >= 5) {
        double acc[5] = {0, 0, 0, 0, 0};
        int cnt[5] = {0, 0, 0, 0, 0};
        for (int i = W::lane(); i < n; i += W::LANES) {
            const double ti = t[i], fi = f[i];
            for (int j = i + 1; j < n; ++j) {
                const double dt = fabs(t[j] - ti);
                const double d = f[j] - fi;
                const double df = d * d;
#pragma unroll
                for (int q = 0; q < 5; ++q)
                    if (dt >= TAU[q] * 0.5 && dt <= TAU[q] * 1.5) { acc[q] += df; ++cnt[q]; }
            }
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const double a = W::sum(acc[q]);
            const int c = W::sum(cnt[q]);
            if (c >= 3) sf[q] = sqrt(a / c);
        }
    }
    double lx[5], ly[5];
    int m = 0;
    for (int q = 0; q < 5; ++q)
        if (!is_nan(sf[q]) && sf[q] > 0) { lx[m] = log10(TAU[q]); ly[m] = log10(sf[q]); ++m; }
    double slope = qnan();
    if (m >= 3) {
        double mx = 0, my = 0;
        for (int i = 0; i < m; ++i) { mx += lx[i]; my += ly[i]; }
        mx /= m; my /= m;
        double sxx = 0, sxy = 0;
        for (int i = 0; i < m; ++i) { sxx += (lx[i] - mx) * (lx[i] - mx); sxy += (lx[i] - mx) * (ly[i] - my); }
        slope = sxy / sxx;
    }
    if (n < 5) slope = qnan();
    if (W::lane() == 0) { for (int q = 0; q < 5; ++q) out6[q] = sf[q]; out6[5] = slope; }
}

// physics_based.py:202-289 on a time-sorted band (lane-0 serial scans over the band; n is small)
template <class W>
LCFE_FN void bazin_simple(const double* t, const double* f, int n, double* out5) {
    if (W::lane() == 0) for (int j = 0; j < 5; ++j) out5[j] = qnan();
    if (n < 5) return;
    const int pk = wave_argmax_first<W>(f, n);
    if (W::lane() != 0) return;
    const double pt = t[pk], pf = f[pk];
    out5[0] = pf;
    out5[1] = pt;
    if (pk + 1 >= 2) {                                          // :235-252
        const double th10 = 0.1 * pf, th90 = 0.9 * pf;
        double t10 = t[0], t90 = pt;
        for (int i = 0; i <= pk; ++i) {
            if (f[i] >= th10 && t10 == t[0]) t10 = t[i];
            if (f[i] >= th90) { t90 = t[i]; break; }
        }
        out5[2] = t90 - t10;
    }
    const int np_ = n - pk;                                     // rows from the peak on
    if (np_ >= 3) {                                             // :258-274
        const double target = pf / 2.718281828459045;
        double fall = qnan();
        for (int i = pk; i < n; ++i)
            if (f[i] <= target) { fall = t[i] - pt; break; }
        if (is_nan(fall) && np_ > 1) fall = (t[n - 1] - pt) * pf / (pf - f[n - 1] + 1e-6);
        out5[3] = fall;
    }
    if (np_ >= 5) {                                             // :277-287
        const int mid = np_ / 2;
        double a = 0, b = 0;
        for (int i = 0; i < mid; ++i) a += f[pk + i];
        for (int i = mid; i < np_; ++i) b += f[pk + i];
        a /= mid;
        b /= (np_ - mid);
        if (a > 0) out5[4] = b / a;
    }
}

template <class W, int CAP>
LCFE_FN void physics_object(const ObjLds<CAP>& L, double z_in, PhysicsLds<CAP>& S) {
    const int lane = W::lane();
    double* o = S.out;
    bool in_bd[6];
    for (int k = 0; k < 6; ++k) in_bd[k] = (L.boff[k + 1] - L.boff[k]) >= 3;     // physics_based.py:306-314
    const int JA[3] = {1, 2, 1}, JB[3] = {2, 3, 3};                              // :318 (g,r) (r,i) (g,i)
    for (int q = 0; q < 3; ++q) {
        const double v = (in_bd[JA[q]] && in_bd[JB[q]]) ? stetson_j<W, CAP>(L, JA[q], JB[q]) : qnan();
        if (lane == 0) o[q] = v;
    }
    for (int q = 0; q < 3; ++q) {                                                // :329-334
        const int k = q + 1, s = L.boff[k], n = L.boff[k + 1] - s;
        const double v = in_bd[k] ? stetson_k<W>(L.bf + s, L.be + s, n) : qnan();
        if (lane == 0) o[3 + q] = v;
    }
    {                                                                            // :338-345
        const int s = L.boff[2], n = L.boff[3] - s;
        structure_function<W>(L.bt + s, L.bf + s, in_bd[2] ? n : 0, o + 6);
    }
    const double z = is_nan(z_in) ? 0.0 : z_in;                                  // :348
    double pkv[6];
    for (int q = 0; q < 3; ++q) {                                                // :351-379
        const int k = q + 1, s = L.boff[k], n = L.boff[k + 1] - s;
        double a = qnan(), b = qnan(), c = qnan();
        pkv[k] = qnan();
        if (in_bd[k]) {
            const double* t = L.bt + s;
            const int pk = wave_argmax_first<W>(L.bf + s, n);
            pkv[k] = L.bf[s + pk];
            a = (t[n - 1] - t[0]) / (1 + z);
            if (pk > 0) b = (t[pk] - t[0]) / (1 + z);
            if (pk < n - 1) c = (t[n - 1] - t[pk]) / (1 + z);
        }
        if (lane == 0) { o[12 + 3 * q] = a; o[13 + 3 * q] = b; o[14 + 3 * q] = c; }
    }
    if (lane == 0) {                                                             // :383-423
        double tp = qnan(), t50 = qnan(), tev = qnan();
        if (in_bd[1] && in_bd[2] && in_bd[3]) {
            tp = estimate_temperature(pkv[1], pkv[2], pkv[3]);
            const int sr = L.boff[2], nr = L.boff[3] - sr;
            int pk = 0;                                                          // argmax of r (first max, NaN first)
            for (int i = 1; i < nr; ++i) {
                const double v = L.bf[sr + i], b = L.bf[sr + pk];
                if (!is_nan(b) && (is_nan(v) || v > b)) pk = i;
            }
            const double target = L.bt[sr + pk] + 50;
            double late[4];
            for (int k = 1; k <= 3; ++k) {
                const int s = L.boff[k], n = L.boff[k + 1] - s;
                const int j = nearest_index(L.bt + s, n, target);
                late[k] = (fabs(L.bt[s + j] - target) < 20) ? L.bf[s + j] : qnan();
            }
            t50 = estimate_temperature(late[1], late[2], late[3]);
            if (!is_nan(tp) && !is_nan(t50)) tev = (t50 - tp) / 50.0;
        }
        o[21] = tp; o[22] = t50; o[23] = tev;
    }
    {                                                                            // :427-433
        const int s = L.boff[2], n = L.boff[3] - s;
        bazin_simple<W>(L.bt + s, L.bf + s, in_bd[2] ? n : 0, o + 24);
    }
    // :437-456 SNR over all rows with err > 0 and flux > 0
    int cnt = 0;
    for (int base = 0; base < L.n; base += W::LANES) {
        const int i = base + lane;
        const bool ok = (i < L.n) && (L.e[i] > 0) && (L.f[i] > 0);
        cnt = wave_compact<W>(ok, ok ? L.f[i] / L.e[i] : 0.0, ok ? L.f[i] : 0.0, S.xs, S.ys, cnt);
    }
    W::sync();
    double msnr = qnan(), medsnr = qnan(), exc = qnan();
    if (cnt > 0) {
        double s = 0;
        for (int i = lane; i < cnt; i += W::LANES) s += S.xs[i];
        msnr = W::sum(s) / cnt;
        medsnr = wave_median<W>(S.xs, cnt, S.slot, S.keys);
        double mf, vf, lo, hi;
        wave_moments<W>(S.ys, cnt, mf, vf, lo, hi);
        double se2 = 0;
        for (int i = lane; i < cnt; i += W::LANES) { const double e = S.ys[i] / S.xs[i]; se2 += e * e; }
        // e = f / (f/e) reproduces err up to rounding; use the staged errors instead for fidelity
        se2 = 0;
        for (int i = lane; i < L.n; i += W::LANES)
            if (L.e[i] > 0 && L.f[i] > 0) se2 += L.e[i] * L.e[i];
        se2 = W::sum(se2) / cnt;
        const double ex = (vf - se2) / (mf * mf);
        exc = (ex > 0) ? ex : 0.0;                                               // Python max(0, x): NaN -> 0
    }
    if (lane == 0) { o[29] = msnr; o[30] = medsnr; o[31] = exc; }
    W::sync();
}

}  // namespace lcfe
