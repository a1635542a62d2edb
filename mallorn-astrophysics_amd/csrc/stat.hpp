// stat.hpp -- per-band + all-band flux statistics (reference: src/features/statistical.py).
//
// 17 statistics for each of u,g,r,i,z,y and for all rows, 3 band-mean ratios and peak_band:
// 123 columns in the order of statistical.py:168-222.
#pragma once
#include "stage.hpp"

namespace lcfe {

constexpr int STAT_NCOL = 123;

// value(s) of rank r in a multiset given per-element strict / non-strict counts: the element
// whose [c_lt, c_le) interval contains r owns it (ties hold equal values, any owner is right).
struct RankSel {
    int r[6];
};

// np.percentile(..., method='linear') interpolation of two neighbours
// (numpy/lib/_function_base_impl.py::_lerp): a + (b-a)*t, or b - (b-a)*(1-t) when t >= 0.5.
LCFE_FN double np_lerp(double a, double b, double t) {
    const double d = b - a;
    return (t >= 0.5) ? (b - d * (1.0 - t)) : (a + d * t);
}

// Statistics of one group (statistical.py:41-132).  gt/gf: the group's rows sorted by time
// (`time_sorted` says whether that holds; if not, neighbours are found by a successor scan);
// ge: the group's errors in the same order.  `sel` is 8 doubles of wave-shared scratch,
// `dev` m doubles of wave-shared scratch.  out17 is wave-shared; lane 0 writes it.
// KPL > 0: the group sorts its fluxes once (LANES x KPL register network, m <= LANES * KPL) and
// reads the order statistics off the sorted copy; KPL == 0: rank counting (any m).
template <class W, int KPL>
LCFE_FN void group_statistics(const double* gt, const double* gf, const double* ge, int m,
                              bool time_sorted, double* sel, double* dev, unsigned long long* keys, double* out17) {
    const int lane = W::lane();
    if (m == 0) {                                    // statistical.py:56-66
        if (lane == 0) {
            out17[0] = 0.0;
            for (int k = 1; k < 17; ++k) out17[k] = qnan();
        }
        return;
    }
    // ---- pass 1: sum, min, max, snr, time extent
    double s = 0.0, mn = __builtin_inf(), mx = -__builtin_inf(), tmn = __builtin_inf(),
           tmx = -__builtin_inf(), snr = 0.0;
    int nsnr = 0;
    bool nanf = false;
    for (int i = lane; i < m; i += W::LANES) {
        const double x = gf[i], tt = gt[i], ee = ge[i];
        s += x;
        nanf = nanf || is_nan(x);
        mn = (x < mn) ? x : mn;
        mx = (x > mx) ? x : mx;
        tmn = (tt < tmn) ? tt : tmn;
        tmx = (tt > tmx) ? tt : tmx;
        if (ee > 0) { snr += fabs(x) / ee; ++nsnr; }
    }
    s = W::sum(s);
    mn = W::min(mn);
    mx = W::max(mx);
    tmn = W::min(tmn);
    tmx = W::max(tmx);
    snr = W::sum(snr);
    nsnr = W::sum(nsnr);
    if (W::any(nanf)) { mn = qnan(); mx = qnan(); }   // np.min/np.max propagate NaN
    const double mean = s / m;
    // ---- pass 2: centred moments (two-pass, as np.std / the reference's skew/kurtosis do)
    double m2 = 0.0;
    for (int i = lane; i < m; i += W::LANES) { const double d = gf[i] - mean; m2 += d * d; }
    m2 = W::sum(m2);
    const double std = (m > 1) ? sqrt(m2 / m) : 0.0;   // statistical.py:71
    double skew = 0.0, kurt = 0.0, b1 = 0.0, b2 = 0.0;
    if (std > 0) {
        double s3 = 0.0, s4 = 0.0;
        int c1 = 0, c2 = 0;
        for (int i = lane; i < m; i += W::LANES) {
            const double x = gf[i];
            const double zz = (x - mean) / std;
            const double z2 = zz * zz;
            s3 += z2 * zz;
            s4 += z2 * z2;
            const double az = fabs(x - mean) / std;      // :91
            c1 += (az > 1.0);
            c2 += (az > 2.0);
        }
        s3 = W::sum(s3);
        s4 = W::sum(s4);
        c1 = W::sum(c1);
        c2 = W::sum(c2);
        if (m > 2) skew = s3 / m;                        // :14-23 (0 if n<3), :77
        if (m > 3) kurt = s4 / m - 3.0;                  // :26-35 (0 if n<4)
        b1 = (double)c1 / m;
        b2 = (double)c2 / m;
    } else if (is_nan(std)) {
        // NaN flux: np.std is NaN; `std > 0` and `std == 0` are both False in the reference, so
        // skewness()/kurtosis() fall through to the NaN-valued formula and beyond_* are 0.
        skew = (m > 2) ? qnan() : 0.0;
        kurt = (m > 3) ? qnan() : ((m > 2) ? 0.0 : 0.0);
    }
    // ---- order statistics by rank counting on sortable keys
    // targets: median lo/hi, p25 lo/hi, p75 lo/hi    (np.median; np.percentile linear)
    const int r_med_lo = (m - 1) / 2, r_med_hi = m / 2;
    const double v25 = 0.25 * (m - 1), v75 = 0.75 * (m - 1);
    const int r25 = (int)floor(v25), r75 = (int)floor(v75);
    const int r25h = (r25 + 1 < m) ? r25 + 1 : m - 1, r75h = (r75 + 1 < m) ? r75 + 1 : m - 1;
    const bool any_nan = W::any(nanf);
    double med, iqr = 0.0, mad;
    if constexpr (KPL > 0) {
        if (any_nan) {
            med = qnan(); mad = qnan();
            if (m > 1) iqr = qnan();
        } else {
            group_sort_values<W, KPL>(gf, m, dev);
            // np.median: mean of the two middle elements
            med = (r_med_lo == r_med_hi) ? dev[r_med_lo] : (dev[r_med_lo] + dev[r_med_hi]) / 2.0;
            if (m > 1) iqr = np_lerp(dev[r75], dev[r75h], v75 - r75) - np_lerp(dev[r25], dev[r25h], v25 - r25);
            // MAD = median(|x - med|); a non-finite median leaves NaN deviations (inf - inf) -> NaN
            mad = (med - med == 0.0) ? mad_of_sorted(dev, m, med) : qnan();
        }
    } else {
        {
            const int ranks[6] = {r_med_lo, r_med_hi, r25, r25h, r75, r75h};
            wave_select_ranks<W, 6>(gf, m, keys, ranks, sel);
        }
        // np.median: mean of the two middle elements; NaN anywhere -> NaN
        med = (r_med_lo == r_med_hi) ? sel[0] : (sel[0] + sel[1]) / 2.0;
        if (m > 1) {
            const double p25 = np_lerp(sel[2], sel[3], v25 - r25);
            const double p75 = np_lerp(sel[4], sel[5], v75 - r75);
            iqr = p75 - p25;
        }
        if (any_nan) { med = qnan(); if (m > 1) iqr = qnan(); }
        W::sync();
        // ---- MAD = median(|x - med|)
        for (int i = lane; i < m; i += W::LANES) dev[i] = fabs(gf[i] - med);
        W::sync();
        {
            const int ranks[2] = {r_med_lo, r_med_hi};
            wave_select_ranks<W, 2>(dev, m, keys, ranks, sel + 6);
        }
        mad = (r_med_lo == r_med_hi) ? sel[6] : (sel[6] + sel[7]) / 2.0;
        if (any_nan || !(med - med == 0.0)) mad = qnan();
    }
    // ---- max slope between time-consecutive rows (statistical.py:99-113)
    double slope = -1.0;    // -1 = "no valid dt" sentinel (slopes are >= 0)
    bool slope_nan = false;
    if (m > 1) {
        for (int i = lane; i < m; i += W::LANES) {
            int nx = -1;
            if (time_sorted) {
                nx = (i + 1 < m) ? i + 1 : -1;
            } else {
                // successor of (t_i, i) in (time, index) order
                const double ti = gt[i];
                double bt = 0.0;
                for (int j = 0; j < m; ++j) {
                    const double tj = gt[j];
                    const bool after = (tj > ti) || (tj == ti && j > i);
                    if (after && (nx < 0 || tj < bt || (tj == bt && j < nx))) { nx = j; bt = tj; }
                }
            }
            if (nx >= 0) {
                const double dt = gt[nx] - gt[i];
                if (dt > 0) {
                    const double sl = fabs((gf[nx] - gf[i]) / dt);
                    if (is_nan(sl)) slope_nan = true;
                    else slope = (sl > slope) ? sl : slope;
                }
            }
        }
        slope = W::max(slope);
        slope_nan = W::any(slope_nan);
    }
    if (lane == 0) {
        out17[0] = (double)m;
        out17[1] = mean;
        out17[2] = std;
        out17[3] = mn;
        out17[4] = mx;
        out17[5] = med;
        out17[6] = skew;
        out17[7] = kurt;
        out17[8] = mx - mn;
        out17[9] = mad;
        out17[10] = iqr;
        out17[11] = b1;
        out17[12] = b2;
        out17[13] = (m > 1) ? (slope_nan ? qnan() : (slope < 0 ? 0.0 : slope)) : 0.0;
        out17[14] = (nsnr > 0) ? snr / nsnr : qnan();           // :116-120
        out17[15] = (m > 1) ? (tmx - tmn) : 0.0;                 // :123-130
        // mean(diff(sort(t))) telescopes exactly: the gaps are exact multiples of one ulp(t)
        out17[16] = (m > 1) ? (tmx - tmn) / (double)(m - 1) : 0.0;
    }
}

// Scratch the statistics kernel needs besides ObjLds.
template <int CAP>
struct StatScratch {
    double dev[CAP];
    unsigned long long keys[CAP];
    double sel[8][8];            // one row per lane group
    double out[STAT_NCOL + 5];
};

// All 123 columns of one staged object into S.out (wave-shared).
// WG: policy of one per-band pass (on the device the six bands run side by side in 8-lane groups of
// the wave); W: policy of the whole wave (the all-rows pass and the cross-band epilogue).
template <class W, class WG, int CAP>
LCFE_FN void stat_object(const ObjLds<CAP>& L, StatScratch<CAP>& S) {
    const int lane = W::lane();
    // sorting-network width of the band groups, uniform over the wave: 4 or 8 values per lane when the
    // longest band fits 8 lanes x that, else rank counting
    int mb = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) { const int c = L.boff[k + 1] - L.boff[k]; mb = (c > mb) ? c : mb; }
    constexpr int GL = (WG::LANES > 1) ? WG::LANES : 8;      // the host build follows the device's choice
    const int per_lane = (mb + GL - 1) / GL;
    for (int k = WG::group_id(); k < 6; k += WG::NGROUPS) {
        const int s = L.boff[k], m = L.boff[k + 1] - s;
        double* o = S.out + 17 * k;
        if (per_lane <= 4)
            group_statistics<WG, 4>(L.bt + s, L.bf + s, L.be + s, m, true, S.sel[WG::group_id()], S.dev + s, S.keys + s, o);
        else if (per_lane <= 8)
            group_statistics<WG, 8>(L.bt + s, L.bf + s, L.be + s, m, true, S.sel[WG::group_id()], S.dev + s, S.keys + s, o);
        else
            group_statistics<WG, 0>(L.bt + s, L.bf + s, L.be + s, m, true, S.sel[WG::group_id()], S.dev + s, S.keys + s, o);
        WG::sync();
    }
    W::sync();
    // all rows on the full wave: CAP / 64 values per lane up to the 512-point tier
    constexpr int KPL_ALL = (W::LANES > 1) ? ((CAP / W::LANES <= 8) ? CAP / W::LANES : 0) : ((CAP <= 512) ? 1 : 0);
    group_statistics<W, KPL_ALL>(L.t, L.f, L.e, L.n, L.sorted != 0, S.sel[0], S.dev, S.keys, S.out + 102);
    W::sync();
    if (lane == 0) {
        double* o = S.out;
        // statistical.py:201-214: ratio of band means, NaN unless numerator is not NaN and denominator > 0
        const double mg = o[17 * 1 + 1], mr = o[17 * 2 + 1], mi = o[17 * 3 + 1], mz = o[17 * 4 + 1];
        o[119] = (!is_nan(mg) && mr > 0) ? mg / mr : qnan();
        o[120] = (!is_nan(mr) && mi > 0) ? mr / mi : qnan();
        o[121] = (!is_nan(mi) && mz > 0) ? mi / mz : qnan();
        // :217-222 first band with the largest max among bands whose max is not NaN
        int pb = -1;
        double best = 0.0;
        for (int k = 0; k < 6; ++k) {
            const double v = o[17 * k + 4];
            if (!is_nan(v) && (pb < 0 || v > best)) { pb = k; best = v; }
        }
        o[122] = (double)pb;
    }
    W::sync();
}

}  // namespace lcfe
