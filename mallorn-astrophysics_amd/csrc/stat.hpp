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

// pass-1 partials of one group (sum, extrema, time extent, SNR sum/count, "has a NaN flux")
struct StatPartial {
    double s, mn, mx, tmn, tmx, snr;
    int nsnr, nan;
};

// np.percentile(..., method='linear') interpolation of two neighbours
// (numpy/lib/_function_base_impl.py::_lerp): a + (b-a)*t, or b - (b-a)*(1-t) when t >= 0.5.
LCFE_FN double np_lerp(double a, double b, double t) {
    const double d = b - a;
    return (t >= 0.5) ? (b - d * (1.0 - t)) : (a + d * t);
}

// lane 0's write of the 17 columns of one group
LCFE_FN void stat_write17(double* out17, int m, double mean, double std, double mn, double mx, double med,
                          double skew, double kurt, double mad, double iqr, double b1, double b2, double slope,
                          bool slope_nan, double snr, int nsnr, double tmn, double tmx) {
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

// a group without rows (statistical.py:56-66): n_obs 0, everything else NaN
LCFE_FN void stat_empty_group(double* out17, StatPartial* part_out) {
    out17[0] = 0.0;
    for (int k = 1; k < 17; ++k) out17[k] = qnan();
    if (part_out) *part_out = StatPartial{0.0, __builtin_inf(), -__builtin_inf(), __builtin_inf(), -__builtin_inf(), 0.0, 0, 0};
}

// band-mean ratios and peak_band from the seven finished groups (one lane)
LCFE_FN void stat_cross_band(double* o) {
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

// |df / dt| for a time step dt > 0 of normal size (the callers discard the other steps).  Device: hardware reciprocal,
// two Newton steps, one product -- within an ulp of the division (max_slope is compared at 1e-9) at a third of its cost.
LCFE_FN double stat_slope(double df, double dt) {
#if defined(__HIP_DEVICE_COMPILE__)
    double inv = __builtin_amdgcn_rcp(dt);
    inv = fma(fma(-dt, inv, 1.0), inv, inv);
    inv = fma(fma(-dt, inv, 1.0), inv, inv);
    return fabs(df * inv);
#else
    return fabs(df / dt);
#endif
}

// Device fast path of group_statistics for a time-sorted group of 1 <= m <= LANES * KPL rows: every
// lane keeps its KPL fluxes (rows lane, lane + LANES, ...) in registers through the three moment
// passes and the sorting network; the loops are unrolled with select-predication (no divergent
// blocks), so the independent per-row chains -- LDS reads, divisions -- overlap.  Same per-lane
// accumulation order as the generic loops below, hence the same sums.
// GATHER (the all-rows group of the lean kernel): the rows are stored band-partitioned and
// `pos_of[i]` is the storage position of file row i, so time-consecutive neighbours are gathered.
// TIME_ENDS (lean kernel): the group's times are known to be ascending and free of NaN, so the time extent
// is read off the first and last row instead of being reduced.
template <class W, int KPL, bool GATHER = false, bool TIME_ENDS = false>
LCFE_FN void group_statistics_fast(const double* gt, const double* gf, const double* ge, int m, double* sorted,
                                   double* out17, StatPartial* part_out, const StatPartial* parts_in,
                                   const unsigned short* pos_of = nullptr) {
    const int lane = W::lane();
    constexpr int PB = (W::LANES == 64) ? 8 : 16;
    (void)PB;
    LCFE_PT0();
    bool ok[KPL];
    int ii[KPL];
    double x[KPL];
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
        const int i = lane + r * W::LANES;
        ok[r] = i < m;
        ii[r] = ok[r] ? i : 0;
        x[r] = gf[ii[r]];
    }
    // ---- pass 1
    double s = 0.0, mn = __builtin_inf(), mx = -__builtin_inf(), tmn = __builtin_inf(),
           tmx = -__builtin_inf(), snr = 0.0;
    int nsnr = 0;
    bool any_nan;
    if (parts_in) {
        int nn = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const StatPartial q = parts_in[k];
            s += q.s;
            snr += q.snr;
            nsnr += q.nsnr;
            nn |= q.nan;
            mn = (q.mn < mn) ? q.mn : mn;
            mx = (q.mx > mx) ? q.mx : mx;
            tmn = (q.tmn < tmn) ? q.tmn : tmn;
            tmx = (q.tmx > tmx) ? q.tmx : tmx;
        }
        any_nan = nn != 0;
    } else {
        bool nanf = false;
#pragma unroll
        for (int r = 0; r < KPL; ++r) {
            const double xr = x[r], ee = ge[ii[r]];
            s += ok[r] ? xr : 0.0;
            nanf = nanf || (ok[r] && is_nan(xr));
            mn = (ok[r] && xr < mn) ? xr : mn;
            mx = (ok[r] && xr > mx) ? xr : mx;
            if constexpr (!TIME_ENDS) {
                const double tt = gt[ii[r]];
                tmn = (ok[r] && tt < tmn) ? tt : tmn;
                tmx = (ok[r] && tt > tmx) ? tt : tmx;
            }
            const bool use = ok[r] && ee > 0;
            const double q = fabs(xr) / ee;
            snr += use ? q : 0.0;
            nsnr += use ? 1 : 0;
        }
        s = W::sum(s);
        mn = W::min(mn);
        mx = W::max(mx);
        if constexpr (TIME_ENDS) { tmn = gt[0]; tmx = gt[m - 1]; }
        else { tmn = W::min(tmn); tmx = W::max(tmx); }
        snr = W::sum(snr);
        nsnr = W::sum(nsnr);
        any_nan = W::any(nanf);
        if (part_out && lane == 0) *part_out = StatPartial{s, mn, mx, tmn, tmx, snr, nsnr, any_nan ? 1 : 0};
    }
    if (any_nan) { mn = qnan(); mx = qnan(); }
    const double mean = s / m;
    LCFE_PT(PB + 0);
    // ---- pass 2
    double m2 = 0.0;
#pragma unroll
    for (int r = 0; r < KPL; ++r) { const double d = x[r] - mean; m2 += ok[r] ? d * d : 0.0; }
    m2 = W::sum(m2);
    const double std = (m > 1) ? sqrt(m2 / m) : 0.0;
    LCFE_PT(PB + 1);
    // ---- pass 3
    double skew = 0.0, kurt = 0.0, b1 = 0.0, b2 = 0.0;
    if (std > 0) {
        double s3 = 0.0, s4 = 0.0;
        int c1 = 0, c2 = 0;
        // z = (x - mean) / std through one reciprocal (1 ulp off the division: the moments keep 1e-15); the
        // counts are exact without it: fl(|d| / std) > k  <=>  |d| > k std for k = 1, 2 (k std is exact, and the
        // next double above k is k (1 + 2^-52), which |d| > k std already reaches before rounding)
        const double inv_std = 1.0 / std, std2 = 2.0 * std;
#pragma unroll
        for (int r = 0; r < KPL; ++r) {
            const double d = x[r] - mean;
            const double zz = d * inv_std;
            const double z2 = zz * zz;
            s3 += ok[r] ? z2 * zz : 0.0;
            s4 += ok[r] ? z2 * z2 : 0.0;
            const double ad = fabs(d);
            c1 += (ok[r] && ad > std) ? 1 : 0;
            c2 += (ok[r] && ad > std2) ? 1 : 0;
        }
        s3 = W::sum(s3);
        s4 = W::sum(s4);
        c1 = W::sum(c1);
        c2 = W::sum(c2);
        if (m > 2) skew = s3 / m;
        if (m > 3) kurt = s4 / m - 3.0;
        b1 = (double)c1 / m;
        b2 = (double)c2 / m;
    } else if (is_nan(std)) {
        skew = (m > 2) ? qnan() : 0.0;
        kurt = (m > 3) ? qnan() : 0.0;
    }
    LCFE_PT(PB + 2);
    // ---- order statistics off one sorted copy
    const int r_med_lo = (m - 1) / 2, r_med_hi = m / 2;
    const double v25 = 0.25 * (m - 1), v75 = 0.75 * (m - 1);
    const int r25 = (int)floor(v25), r75 = (int)floor(v75);
    const int r25h = (r25 + 1 < m) ? r25 + 1 : m - 1, r75h = (r75 + 1 < m) ? r75 + 1 : m - 1;
    double med = qnan(), iqr = (m > 1) ? qnan() : 0.0, mad = qnan();
    if (!any_nan) {
        double v[KPL];
#pragma unroll
        for (int r = 0; r < KPL; ++r) v[r] = ok[r] ? x[r] : __builtin_inf();
        sort_merges<W, KPL, 2>(v);
        const int base = lane * KPL;
#pragma unroll
        for (int r = 0; r < KPL; ++r)
            if (base + r < m) sorted[base + r] = v[r];
        W::sync();
        LCFE_PT(PB + 3);
        const double a0 = sorted[r_med_lo], a1 = sorted[r_med_hi], q0 = sorted[r25], q1 = sorted[r25h],
                     q2 = sorted[r75], q3 = sorted[r75h];
        med = (r_med_lo == r_med_hi) ? a0 : (a0 + a1) / 2.0;
        LCFE_PT(PB + 8);
        if (m > 1) iqr = np_lerp(q2, q3, v75 - r75) - np_lerp(q0, q1, v25 - r25);
        LCFE_PT(PB + 9);
        mad = (med - med == 0.0) ? mad_of_sorted<W>(sorted, m, med) : qnan();
        LCFE_PT(PB + 10);
    }
    LCFE_PT(PB + 4);
    // ---- max slope between time-consecutive rows
    double slope = -1.0;
    bool slope_nan = false;
    if (m > 1) {
#pragma unroll
        for (int r = 0; r < KPL; ++r) {
            const int i = lane + r * W::LANES;
            const bool has = i + 1 < m;
            const int nx = has ? i + 1 : 0;
            double dt, df;
            if constexpr (GATHER) {
                const int p0 = pos_of[ii[r]], p1 = pos_of[nx];
                dt = gt[p1] - gt[p0];
                df = gf[p1] - gf[p0];
            } else {
                dt = gt[nx] - gt[ii[r]];
                df = gf[nx] - x[r];
            }
            const double sl = stat_slope(df, dt);
            const bool valid = has && dt > 0;
            slope_nan = slope_nan || (valid && is_nan(sl));
            slope = (valid && sl > slope) ? sl : slope;
        }
        slope = W::max(slope);
        slope_nan = W::any(slope_nan);
    }
    LCFE_PT(PB + 5);
    if (lane == 0)
        stat_write17(out17, m, mean, std, mn, mx, med, skew, kurt, mad, iqr, b1, b2, slope, slope_nan, snr, nsnr, tmn, tmx);
    LCFE_PT(PB + 6);
}

// Statistics of one group (statistical.py:41-132).  gt/gf: the group's rows sorted by time
// (`time_sorted` says whether that holds; if not, neighbours are found by a successor scan);
// ge: the group's errors in the same order.  `sel` is 8 doubles of wave-shared scratch,
// `dev` m doubles of wave-shared scratch.  out17 is wave-shared; lane 0 writes it.
// KPL > 0: the group sorts its fluxes once (LANES x KPL register network, m <= LANES * KPL) and
// reads the order statistics off the sorted copy; KPL == 0: rank counting (any m).
// `part_out` (may be null): the group's pass-1 partials, for the all-rows pass to combine;
// `parts_in` (may be null): six band partials that together cover exactly this group's rows.
template <class W, int KPL>
LCFE_FN void group_statistics(const double* gt, const double* gf, const double* ge, int m,
                              bool time_sorted, double* sel, double* dev, unsigned long long* keys, double* out17,
                              StatPartial* part_out, const StatPartial* parts_in) {
    const int lane = W::lane();
    if (m == 0) {                                    // statistical.py:56-66
        if (lane == 0) stat_empty_group(out17, part_out);
        return;
    }
    constexpr int PB = (W::LANES == 64) ? 8 : 16;   // phase-profile slots (debug builds)
    (void)PB;
    LCFE_PT0();
    if constexpr (KPL > 0 && W::LANES > 1) {
        if (time_sorted) {
            group_statistics_fast<W, KPL>(gt, gf, ge, m, dev, out17, part_out, parts_in);
            return;
        }
    }
    // ---- pass 1: sum, min, max, snr, time extent
    double s = 0.0, mn = __builtin_inf(), mx = -__builtin_inf(), tmn = __builtin_inf(),
           tmx = -__builtin_inf(), snr = 0.0;
    int nsnr = 0;
    bool any_nan;
    if (parts_in) {
        int nn = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const StatPartial q = parts_in[k];
            s += q.s;
            snr += q.snr;
            nsnr += q.nsnr;
            nn |= q.nan;
            mn = (q.mn < mn) ? q.mn : mn;
            mx = (q.mx > mx) ? q.mx : mx;
            tmn = (q.tmn < tmn) ? q.tmn : tmn;
            tmx = (q.tmx > tmx) ? q.tmx : tmx;
        }
        any_nan = nn != 0;
    } else {
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
        any_nan = W::any(nanf);
        if (part_out && lane == 0) *part_out = StatPartial{s, mn, mx, tmn, tmx, snr, nsnr, any_nan ? 1 : 0};
    }
    if (any_nan) { mn = qnan(); mx = qnan(); }   // np.min/np.max propagate NaN
    const double mean = s / m;
    LCFE_PT(PB + 0);
    // ---- pass 2: centred moments (two-pass, as np.std / the reference's skew/kurtosis do)
    double m2 = 0.0;
    for (int i = lane; i < m; i += W::LANES) { const double d = gf[i] - mean; m2 += d * d; }
    m2 = W::sum(m2);
    const double std = (m > 1) ? sqrt(m2 / m) : 0.0;   // statistical.py:71
    LCFE_PT(PB + 1);
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
            const double az = fabs(zz);                  // :91 |x - mean| / std (division is sign-symmetric)
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
    LCFE_PT(PB + 2);
    // ---- order statistics by rank counting on sortable keys
    // targets: median lo/hi, p25 lo/hi, p75 lo/hi    (np.median; np.percentile linear)
    const int r_med_lo = (m - 1) / 2, r_med_hi = m / 2;
    const double v25 = 0.25 * (m - 1), v75 = 0.75 * (m - 1);
    const int r25 = (int)floor(v25), r75 = (int)floor(v75);
    const int r25h = (r25 + 1 < m) ? r25 + 1 : m - 1, r75h = (r75 + 1 < m) ? r75 + 1 : m - 1;
    double med, iqr = 0.0, mad;
    if constexpr (KPL > 0) {
        if (any_nan) {
            med = qnan(); mad = qnan();
            if (m > 1) iqr = qnan();
        } else {
            group_sort_values<W, KPL>(gf, m, dev);
            LCFE_PT(PB + 3);
            // np.median: mean of the two middle elements
            med = (r_med_lo == r_med_hi) ? dev[r_med_lo] : (dev[r_med_lo] + dev[r_med_hi]) / 2.0;
            if (m > 1) iqr = np_lerp(dev[r75], dev[r75h], v75 - r75) - np_lerp(dev[r25], dev[r25h], v25 - r25);
            // MAD = median(|x - med|); a non-finite median leaves NaN deviations (inf - inf) -> NaN
            mad = (med - med == 0.0) ? mad_of_sorted<W>(dev, m, med) : qnan();
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
    LCFE_PT(PB + 4);
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
    LCFE_PT(PB + 5);
    if (lane == 0)
        stat_write17(out17, m, mean, std, mn, mx, med, skew, kurt, mad, iqr, b1, b2, slope, slope_nan, snr, nsnr, tmn, tmx);
    LCFE_PT(PB + 6);
}

// Scratch the statistics kernel needs besides ObjLds.
template <int CAP>
struct StatScratch {
    double dev[CAP];
    unsigned long long keys[CAP];
    double sel[8][8];            // one row per lane group
    double out[STAT_NCOL + 5];
    StatPartial part[6];
};

// All 123 columns of one staged object into S.out (wave-shared).
// WG: policy of one per-band pass (on the device the six bands run side by side in 8-lane groups of
// the wave); W: policy of the whole wave (the all-rows pass and the cross-band epilogue).
template <class W, class WG, int CAP>
LCFE_FN void stat_object(const ObjLds<CAP>& L, StatScratch<CAP>& S) {
    const int lane = W::lane();
    LCFE_PT0();
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
            group_statistics<WG, 4>(L.bt + s, L.bf + s, L.be + s, m, true, S.sel[WG::group_id()], S.dev + s, S.keys + s, o, &S.part[k], nullptr);
        else if (per_lane <= 8)
            group_statistics<WG, 8>(L.bt + s, L.bf + s, L.be + s, m, true, S.sel[WG::group_id()], S.dev + s, S.keys + s, o, &S.part[k], nullptr);
        else
            group_statistics<WG, 0>(L.bt + s, L.bf + s, L.be + s, m, true, S.sel[WG::group_id()], S.dev + s, S.keys + s, o, &S.part[k], nullptr);
        WG::sync();
    }
    W::sync();
    LCFE_PT(1);
    // all rows on the full wave: CAP / 64 values per lane up to the 512-point tier
    constexpr int KPL_ALL = (W::LANES > 1) ? ((CAP / W::LANES <= 8) ? CAP / W::LANES : 0) : ((CAP <= 512) ? 1 : 0);
    // the band partials cover all rows unless some row has an unknown band code
    group_statistics<W, KPL_ALL>(L.t, L.f, L.e, L.n, L.sorted != 0, S.sel[0], S.dev, S.keys, S.out + 102, nullptr,
                                 (L.boff[6] == L.n) ? S.part : nullptr);
    LCFE_PT(2);
    W::sync();
    if (lane == 0) stat_cross_band(S.out);
    W::sync();
    LCFE_PT(4);
}

}  // namespace lcfe
