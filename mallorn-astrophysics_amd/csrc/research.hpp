// research.hpp -- the v115 "research" features (reference: src/features/research_features.py) -> 40 columns:
// log-log power-law decay quality per band (:44-160), nuclear-position proxy (:163-247), colour at peak
// (:250-331), Mexican-hat power spectra of the r band (:338-430), luminosity features (:437-530).
//
// Inputs are the LDS views of stage.hpp: band-partitioned time-sorted copies where the reference sorts
// (``sort_values('Time (MJD)')``: :126, :183, :377, :494) and the file-order arrays where it does not (colour at
// peak: :273-300).  np.polyfit(x, y, 1) is the closed-form least-squares line on centred sums (tde.hpp),
// np.percentile / np.median are rank selections, scipy.signal.convolve(mode='same') is the direct sum.
#pragma once
#include "fits.hpp"      // wave_median
#include "stage.hpp"
#include "tde.hpp"       // wave_linfit, wave_compact

namespace lcfe {

constexpr int RESEARCH_NCOL = 40;
// longest 1-day grid of the r band the MHPS pass takes (days of time span): 32 KiB of LDS in the LDS tiers, 512 KiB of
// global scratch in the long-object tier (CAP > 2048)
template <int CAP> constexpr int research_grid() { return (CAP > 2048) ? 65536 : 4096; }
constexpr int RESEARCH_WAVELET = 500;     // 5 x the largest scale (100 d)

template <int CAP>
struct ResearchLds {
    double xs[CAP], ys[CAP], zs[CAP];
    unsigned long long keys[CAP];
    double grid[research_grid<CAP>()];
    double wv[RESEARCH_WAVELET];
    double slot[4];
    double out[RESEARCH_NCOL];
};

LCFE_FN double np_clip_nan(double x, double lo, double hi) {       // np.clip keeps NaN
    return (x != x) ? x : ((x < lo) ? lo : ((x > hi) ? hi : x));
}

// research_features.py:44-117 for one time-sorted band of m >= 5 rows -> o6
template <class W, int CAP>
LCFE_FN void research_power_law_band(const double* t, const double* f, const double* e, int m, ResearchLds<CAP>& S, double* o6) {
    const int lane = W::lane();
    const int pk = wave_argmax_first<W>(f, m);                      // :70 np.argmax
    const double peak_time = t[pk];
    int cnt = 0;
    for (int base = 0; base < m; base += W::LANES) {                // :75-78
        const int i = base + lane;
        const bool in = i < m;
        const double ti = in ? t[i] : 0.0, fi = in ? f[i] : 0.0, ei = in ? e[i] : 0.0;
        const bool sel = in && (ti > peak_time + 10) && (fi > 0);
        const unsigned long long mask = W::ballot(sel);
        if (sel) {
            const int pos = cnt + W::prefix(mask);
            S.xs[pos] = log10(ti - peak_time);                      // :84-86
            S.ys[pos] = log10(fi);
            S.zs[pos] = np_clip_nan(ei / (fi * log(10.0) + 1e-10), 0.01, 1.0);   // :105-106
        }
        cnt += popcll(mask);
    }
    W::sync();
    if (cnt < 4) {                                                  // :80
        if (lane == 0) { for (int k = 0; k < 5; ++k) o6[k] = qnan(); o6[5] = 0.0; }
        return;
    }
    double slope, icpt;
    wave_linfit<W>(S.xs, S.ys, cnt, slope, icpt);                   // :90 (a singular design makes polyfit raise: NaN, success 0)
    if (is_nan(slope)) {
        if (lane == 0) { for (int k = 0; k < 5; ++k) o6[k] = qnan(); o6[5] = 0.0; }
        return;
    }
    double sr = 0;
    for (int i = lane; i < cnt; i += W::LANES) sr += S.ys[i] - (slope * S.xs[i] + icpt);
    const double rmean = W::sum(sr) / cnt;
    double q = 0, c2 = 0;
    for (int i = lane; i < cnt; i += W::LANES) {
        const double r = S.ys[i] - (slope * S.xs[i] + icpt);       // :98-99
        const double d = r - rmean;
        q += d * d;
        const double u = r / S.zs[i];
        c2 += u * u;
    }
    q = W::sum(q);
    c2 = W::sum(c2);
    if (lane == 0) {
        o6[0] = slope;
        o6[1] = fabs(slope - (-5.0 / 3.0));
        o6[2] = fabs(slope - (-5.0 / 12.0));
        o6[3] = c2 / (double)((cnt - 2 > 1) ? cnt - 2 : 1);         // :107-109
        o6[4] = sqrt(q / cnt);                                      // :100 np.std
        o6[5] = 1.0;
    }
    W::sync();
}

// linear-interpolated percentile p (0..1) of m wave-shared values (np.percentile, method 'linear'); NaN if any NaN
template <class W, int CAP>
LCFE_FN double research_percentile(const double* x, int m, double p, ResearchLds<CAP>& S) {
    bool nanf = false;
    for (int i = W::lane(); i < m; i += W::LANES) nanf = nanf || is_nan(x[i]);
    const double v = p * (m - 1);
    const int lo = (int)floor(v), hi = (lo + 1 < m) ? lo + 1 : m - 1;
    const int ranks[2] = {lo, hi};
    wave_select_ranks<W, 2>(x, m, S.keys, ranks, S.slot);
    const double a = S.slot[0], b = S.slot[1];
    const bool any_nan = W::any(nanf);
    W::sync();
    return any_nan ? qnan() : np_lerp(a, b, v - lo);
}

// research_features.py:163-247 on the time-sorted r band -> o4
template <class W, int CAP>
LCFE_FN void research_nuclear(const double* t, const double* f, const double* e, int m, ResearchLds<CAP>& S, double* o4) {
    const int lane = W::lane();
    double sm = qnan(), conc = qnan(), ratio = qnan(), score = qnan();
    if (m >= 10) {                                                  // :185
        for (int i = lane; i + 1 < m; i += W::LANES)                // :193-195
            S.xs[i] = fabs(f[i + 1] - f[i]) / ((t[i + 1] - t[i]) + 0.1);
        W::sync();
        const double med_rate = wave_median<W>(S.xs, m - 1, S.slot, S.keys);
        const double med_err = wave_median<W>(e, m, S.slot, S.keys);
        if (med_err > 0) sm = 1.0 / (1.0 + med_rate / med_err);    // :199-201
        double mean, var, mn, mx;
        wave_moments<W>(f, m, mean, var, mn, mx);
        bool nanf = false;
        for (int i = lane; i < m; i += W::LANES) nanf = nanf || is_nan(f[i]);
        const double peak = W::any(nanf) ? qnan() : mx;             // np.max propagates NaN
        const double base = research_percentile<W, CAP>(f, m, 0.10, S);   // :206
        if (base > 0) conc = peak / base;
        else if (peak > 0) {                                        // :210-211
            for (int i = lane; i < m; i += W::LANES) S.xs[i] = fabs(f[i]) + 1.0;
            W::sync();
            conc = peak / wave_median<W>(S.xs, m, S.slot, S.keys);
        }
        if (m >= 20) {                                              // :215-227
            double ssum = 0;
            int scount = 0;
            for (int i = lane; i < m - 5; i += W::LANES) {
                if (t[i + 5] - t[i] < 15) {
                    double s5 = 0;
                    for (int k = 0; k < 5; ++k) s5 += f[i + k];
                    const double m5 = s5 / 5.0;
                    double q5 = 0;
                    for (int k = 0; k < 5; ++k) { const double d = f[i + k] - m5; q5 += d * d; }
                    ssum += sqrt(q5 / 5.0);
                    ++scount;
                }
            }
            ssum = W::sum(ssum);
            scount = W::sum(scount);
            const double long_var = sqrt(var);
            if (scount > 0 && long_var > 0) ratio = (ssum / scount) / long_var;
        }
        double acc = 0;                                             // :230-243
        int ns = 0;
        if (!is_nan(sm)) { acc += sm; ++ns; }
        if (!is_nan(conc)) { const double c = conc / 100; acc += (c < 1.0) ? c : 1.0; ++ns; }      // python min(1.0, x)
        if (!is_nan(ratio)) { acc += 1.0 - ((ratio < 1.0) ? ratio : 1.0); ++ns; }
        if (ns > 0) score = acc / ns;
    }
    if (lane == 0) { o4[0] = sm; o4[1] = conc; o4[2] = ratio; o4[3] = score; }
    W::sync();
}

// time of the first maximum of the FILE-ORDER rows of band k, NaN fluxes skipped (pandas idxmax); count -> n
template <class W, int CAP>
LCFE_FN double research_band_peak_time(const ObjLds<CAP>& L, int k, int& n) {
    double best = -__builtin_inf();
    int c = 0;
    for (int i = W::lane(); i < L.n; i += W::LANES)
        if (L.b[i] == k) { ++c; if (!is_nan(L.f[i])) best = fmax(best, L.f[i]); }
    n = W::sum(c);
    best = W::max(best);
    int cand = 0x7fffffff;
    for (int i = W::lane(); i < L.n; i += W::LANES)
        if (L.b[i] == k && L.f[i] == best) cand = (i < cand) ? i : cand;
    cand = W::min(cand);
    return (cand == 0x7fffffff) ? qnan() : L.t[cand];
}

// flux of the file-order row of band k nearest to `pt` among those with |t - pt| < 10 (first minimum); found -> ok
template <class W, int CAP>
LCFE_FN double research_nearest_flux(const ObjLds<CAP>& L, int k, double pt, bool& ok) {
    double bd = __builtin_inf();
    for (int i = W::lane(); i < L.n; i += W::LANES)
        if (L.b[i] == k) { const double d = fabs(L.t[i] - pt); if (d < 10) bd = fmin(bd, d); }
    bd = W::min(bd);
    int cand = 0x7fffffff;
    for (int i = W::lane(); i < L.n; i += W::LANES)
        if (L.b[i] == k && fabs(L.t[i] - pt) == bd && bd < 10) cand = (i < cand) ? i : cand;
    cand = W::min(cand);
    ok = cand != 0x7fffffff;
    return ok ? L.f[cand] : qnan();
}

// research_features.py:250-331 -> o4 (file-order rows)
template <class W, int CAP>
LCFE_FN void research_color_at_peak(const ObjLds<CAP>& L, double* o4) {
    const int lane = W::lane();
    double res[4] = {qnan(), qnan(), qnan(), qnan()};
    int nr = 0, ng = 0;
    double peak_time = research_band_peak_time<W, CAP>(L, 2, nr);  // :272-279
    bool have = nr >= 3;
    if (!have) {
        peak_time = research_band_peak_time<W, CAP>(L, 1, ng);
        have = ng >= 3;
    }
    if (have && !is_nan(peak_time)) {
        const int K1[2] = {1, 2}, K2[2] = {2, 3};
        for (int p = 0; p < 2; ++p) {
            const int k1 = K1[p], k2 = K2[p];
            int c1 = 0, c2 = 0;
            for (int i = lane; i < L.n; i += W::LANES) { c1 += (L.b[i] == k1); c2 += (L.b[i] == k2); }
            c1 = W::sum(c1);
            c2 = W::sum(c2);
            if (c1 < 2 || c2 < 2) continue;                         // :285
            bool ok1, ok2;
            const double f1 = research_nearest_flux<W, CAP>(L, k1, peak_time, ok1);    // :291-300
            const double f2 = research_nearest_flux<W, CAP>(L, k2, peak_time, ok2);
            if (!(ok1 && ok2 && f1 > 0 && f2 > 0)) continue;
            const double cpk = -2.5 * log10(f1 / f2);               // :304
            res[2 * p] = cpk;
            // late colours (:308-329): every late band-1 row against its nearest late band-2 row (first minimum)
            double acc = 0;
            int cnt = 0, late2 = 0;
            for (int i = lane; i < L.n; i += W::LANES) late2 += (L.b[i] == k2 && L.t[i] > peak_time + 50);
            late2 = W::sum(late2);
            if (late2 > 0) {
                for (int i = lane; i < L.n; i += W::LANES) {
                    if (!(L.b[i] == k1 && L.t[i] > peak_time + 50)) continue;
                    double bd = __builtin_inf();
                    int bj = -1;
                    for (int j = 0; j < L.n; ++j) {
                        if (!(L.b[j] == k2 && L.t[j] > peak_time + 50)) continue;
                        const double d = fabs(L.t[j] - L.t[i]);
                        if (bj < 0 || d < bd) { bd = d; bj = j; }   // NaN distances never win: np.argmin would pick the NaN, times are finite by contract
                    }
                    if (bj >= 0 && bd < 5) {
                        const double fa = L.f[i], fb = L.f[bj];
                        if (fa > 0 && fb > 0) { acc += -2.5 * log10(fa / fb); ++cnt; }
                    }
                }
                acc = W::sum(acc);
                cnt = W::sum(cnt);
                if (cnt > 0) res[2 * p + 1] = acc / cnt - cpk;      // :329
            }
        }
    }
    if (lane == 0) { o4[0] = res[0]; o4[1] = res[1]; o4[2] = res[2]; o4[3] = res[3]; }
    W::sync();
}

// research_features.py:352-430 on the time-sorted r band -> o6; returns false if the 1-day grid does not fit
template <class W, int CAP>
LCFE_FN bool research_mhps(const double* t, const double* f, int m, ResearchLds<CAP>& S, double* o6) {
    const int lane = W::lane();
    if (lane == 0) for (int k = 0; k < 6; ++k) o6[k] = qnan();
    W::sync();
    if (m < 20) return true;                                        // :379
    const double span = t[m - 1] - t[0];
    if (!(span >= 50)) return true;                                 // :387 (NaN span: `time_span < 50` is False in the reference, but its arange then raises)
    const double nd = ceil(span);                                   // len(np.arange(t0, t_last, 1.0))
    if (!(nd <= (double)research_grid<CAP>())) return false;
    const int N = (int)nd;
    // np.interp on the regular grid (:391-394): x_k = t0 + k
    // np.arange fills start, start + step, then start + k * delta with delta = (start + step) - start -- not exactly 1
    const double second = t[0] + 1.0;
    const double delta = second - t[0];
    double ssum = 0;
    for (int k = lane; k < N; k += W::LANES) {
        const double x = (k == 0) ? t[0] : ((k == 1) ? second : t[0] + (double)k * delta);
        int lo = 0, hi = m - 1;                                     // last j with t[j] <= x
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (t[mid] <= x) lo = mid; else hi = mid - 1; }
        double v;
        if (lo >= m - 1) v = f[m - 1];
        else if (t[lo] == x) v = f[lo];
        else {
            const double slope = (f[lo + 1] - f[lo]) / (t[lo + 1] - t[lo]);
            v = slope * (x - t[lo]) + f[lo];
            if (is_nan(v)) {                                        // numpy's fallback branches
                v = slope * (x - t[lo + 1]) + f[lo + 1];
                if (is_nan(v) && f[lo] == f[lo + 1]) v = f[lo];
            }
        }
        S.grid[k] = v;
        ssum += v;
    }
    const double mean = W::sum(ssum) / N;                           // :397
    W::sync();
    for (int k = lane; k < N; k += W::LANES) S.grid[k] -= mean;
    W::sync();
    const double scales[3] = {10.0, 30.0, 100.0};
    double pw[3] = {qnan(), qnan(), qnan()};
    bool got[3] = {false, false, false};
    for (int s = 0; s < 3; ++s) {
        const double sc = scales[s];
        int Lw = (int)fmin(5.0 * sc, (double)(N / 2));              // :403
        if (Lw < 5) continue;
        // mexican_hat_wavelet (:338-349): t = np.linspace(-(Lw // 2 rounded down), Lw // 2, Lw)
        const double start = -(double)((Lw + 1) / 2), stop = (double)(Lw / 2);   // python: -Lw // 2 == -ceil(Lw / 2)
        const double step = (stop - start) / (double)(Lw - 1);
        double e2 = 0;
        for (int k = lane; k < Lw; k += W::LANES) {
            const double tk = (k == Lw - 1) ? stop : (double)k * step + start;
            const double x = tk / sc;
            const double w = (1.0 - x * x) * exp(-(x * x) / 2.0);
            S.wv[k] = w;
            e2 += w * w;
        }
        const double norm = sqrt(W::sum(e2));
        W::sync();
        for (int k = lane; k < Lw; k += W::LANES) S.wv[k] /= norm;
        W::sync();
        // scipy.signal.convolve(f, w, mode='same'): out[i] = sum_j f[j] w[i + h - j], h = (Lw - 1) // 2
        // Four consecutive outputs per lane: at step d the lane's outputs i0 + c need f[i0 + c + h - (Lw - 1) + d] and all
        // lanes the same w[Lw - 1 - d], so one new grid value and one (broadcast) wavelet value serve four multiply-adds
        // -- a quarter of the LDS reads of one output per lane.  Every output still adds its terms in ascending j; terms
        // outside the grid enter as 0 * w, which leaves a sum unchanged (at most the sign of a zero, and the output is squared).
        const int h = (Lw - 1) / 2;
        double p2 = 0;
        for (int base = 0; base < N; base += 4 * W::LANES) {
            const int i0 = base + 4 * lane;
            const int jb = i0 + h - (Lw - 1);                       // f index of output i0 at step 0
            auto g_at = [&](int j) { return (j >= 0 && j < N) ? S.grid[j] : 0.0; };
            double g0 = g_at(jb), g1 = g_at(jb + 1), g2 = g_at(jb + 2), g3 = g_at(jb + 3);
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            for (int d = 0; d < Lw; ++d) {
                const double wk = S.wv[Lw - 1 - d];
                a0 += g0 * wk; a1 += g1 * wk; a2 += g2 * wk; a3 += g3 * wk;
                g0 = g1; g1 = g2; g2 = g3; g3 = g_at(jb + d + 4);
            }
            if (i0 < N) p2 += a0 * a0;
            if (i0 + 1 < N) p2 += a1 * a1;
            if (i0 + 2 < N) p2 += a2 * a2;
            if (i0 + 3 < N) p2 += a3 * a3;
        }
        pw[s] = W::sum(p2) / N;                                     // :413
        got[s] = true;
        W::sync();
    }
    if (lane == 0) {
        o6[0] = pw[0]; o6[1] = pw[1]; o6[2] = pw[2];
        if (got[0] && got[2] && pw[2] > 0) o6[3] = pw[0] / pw[2];   // :419-423
        if (got[1] && got[2] && pw[2] > 0) o6[4] = pw[1] / pw[2];
        int best = -1;                                              // :426-428 python max(dict, key=get): first maximum
        for (int s = 0; s < 3; ++s)
            if (got[s] && (best < 0 || pw[s] > pw[best])) best = s;
        if (best >= 0) o6[5] = scales[best];
    }
    W::sync();
    return true;
}

// research_features.py:437-530 (+ the Z > 0 guard of :552-559) -> o5
template <class W, int CAP>
LCFE_FN void research_luminosity(const ObjLds<CAP>& L, double z, ResearchLds<CAP>& S, double* o5) {
    const int lane = W::lane();
    if (lane == 0) for (int k = 0; k < 5; ++k) o5[k] = qnan();
    W::sync();
    if (!(z > 0)) return;                                           // :553 (NaN redshift: no luminosity features)
    const double c_h0 = 299792.458 / 70.0;
    double d_l;
    if (z < 0.1) d_l = c_h0 * z * (1 + z / 2);                       // :449-451
    else { const double q0 = 0.5 * 0.3 - 0.7; d_l = c_h0 * z * (1 + 0.5 * (1 - q0) * z); }   // :455-456
    if (lane == 0) o5[0] = d_l;
    // g, r, i rows: one contiguous range of the band-partitioned copy (segments in u,g,r,i,z,y order)
    const int s0 = L.boff[1], mo = L.boff[4] - s0;
    if (mo < 5) { W::sync(); return; }                              // :490
    const double d2 = d_l * d_l;
    for (int i = lane; i < mo; i += W::LANES) S.xs[i] = L.bf[s0 + i] * d2;    // :500
    W::sync();
    double mean, var, mn, mx;
    wave_moments<W>(S.xs, mo, mean, var, mn, mx);
    bool nanf = false;
    for (int i = lane; i < mo; i += W::LANES) nanf = nanf || is_nan(S.xs[i]);
    const bool any_nan = W::any(nanf);
    const double peak = any_nan ? qnan() : mx;
    const double base = research_percentile<W, CAP>(S.xs, mo, 0.10, S);
    if (lane == 0) { o5[1] = peak; o5[2] = peak - base; o5[3] = mean; }
    // np.argmax of the time-sorted luminosities: the first maximum in (time, file index) order; a NaN is the maximum
    double bt = __builtin_inf();
    int bi = 0x7fffffff;
    for (int i = lane; i < mo; i += W::LANES) {
        const double v = S.xs[i];
        const bool is_max = any_nan ? is_nan(v) : (v == mx);
        if (is_max) {
            const double ti = L.bt[s0 + i];
            const int fi = L.bidx[s0 + i];
            if (ti < bt || (ti == bt && fi < bi)) { bt = ti; bi = fi; }
        }
    }
    const double tpk = W::min(bt);
    int cand = (bt == tpk) ? bi : 0x7fffffff;
    const int ipk = W::min(cand);
    // rows at or after the peak in sorted order (:513-516): n_post = len - peak_idx
    int cnt = 0;
    double lmin = __builtin_inf();
    bool post_nan = false;
    for (int base_i = 0; base_i < mo; base_i += W::LANES) {
        const int i = base_i + lane;
        const bool in = i < mo;
        const double ti = in ? L.bt[s0 + i] : 0.0;
        const int fi = in ? (int)L.bidx[s0 + i] : 0;
        const bool sel = in && (ti > tpk || (ti == tpk && fi >= ipk));
        const double lum = sel ? S.xs[i] : 1.0;
        if (sel) { lmin = fmin(lmin, lum); post_nan = post_nan || is_nan(lum); }
        cnt = wave_compact<W>(sel, ti - tpk, log10(lum), S.ys, S.zs, cnt);
    }
    W::sync();
    lmin = W::min(lmin);
    if (W::any(post_nan)) lmin = qnan();                            // np.min propagates NaN
    if (cnt > 5 && cnt >= 3 && lmin > 0) {                          // :514, :519
        double m2, v2, a2, b2;
        wave_moments<W>(S.ys, cnt, m2, v2, a2, b2);
        if (sqrt(v2) > 0) {                                         // :523 np.std(dt) > 0
            double slope, icpt;
            wave_linfit<W>(S.ys, S.zs, cnt, slope, icpt);
            if (lane == 0) o5[4] = slope * 100;                     // :525-526
        }
    }
    W::sync();
}

// All 40 columns of one staged object into S.out; returns the status word (0, or -100: r-band span beyond the MHPS grid)
template <class W, int CAP>
LCFE_FN int research_object(const ObjLds<CAP>& L, double z, ResearchLds<CAP>& S) {
    const int lane = W::lane();
    double* o = S.out;
    // 1. power law per band g, r, i + the optical summary (:120-160)
    for (int k = 1; k <= 3; ++k) {
        const int s = L.boff[k], m = L.boff[k + 1] - s;
        double* o6 = o + 6 * (k - 1);
        if (m < 5) { if (lane == 0) for (int q = 0; q < 6; ++q) o6[q] = qnan(); }   // :128-133
        else research_power_law_band<W, CAP>(L.bt + s, L.bf + s, L.be + s, m, S, o6);
        W::sync();
    }
    if (lane == 0) {
        double a[3];
        int na = 0;
        for (int k = 0; k < 3; ++k) if (!is_nan(o[6 * k])) a[na++] = o[6 * k];
        if (na >= 2) {                                              // :147-152
            double s = 0, d = 0;
            for (int k = 0; k < na; ++k) { s += a[k]; d += fabs(a[k] - (-5.0 / 3.0)); }
            const double mean = s / na;
            double q = 0;
            for (int k = 0; k < na; ++k) q += (a[k] - mean) * (a[k] - mean);
            o[18] = mean; o[19] = sqrt(q / na); o[20] = d / na;
        } else {                                                    // :153-158
            o[18] = na ? a[0] : qnan();
            o[19] = qnan();
            o[20] = na ? fabs(a[0] - (-5.0 / 3.0)) : qnan();
        }
    }
    W::sync();
    // 2. nuclear-position proxy (r band)
    {
        const int s = L.boff[2], m = L.boff[3] - s;
        research_nuclear<W, CAP>(L.bt + s, L.bf + s, L.be + s, m, S, o + 21);
    }
    // 3. colour at peak
    research_color_at_peak<W, CAP>(L, o + 25);
    // 4. Mexican-hat power spectra (r band)
    bool fits_grid;
    {
        const int s = L.boff[2], m = L.boff[3] - s;
        fits_grid = research_mhps<W, CAP>(L.bt + s, L.bf + s, m, S, o + 29);
    }
    // 5. luminosity
    research_luminosity<W, CAP>(L, z, S, o + 35);
    return fits_grid ? 0 : -100;
}

}  // namespace lcfe
