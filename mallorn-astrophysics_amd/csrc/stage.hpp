// stage.hpp -- bring one object's CSR slice into LDS and derive the views every extractor needs.
//
// The reference does, per object and per band, ``obj_lc[obj_lc['Filter'] == band]
// .sort_values('Time (MJD)')`` (bazin_fitting.py:195, tde_physics.py:39, lightcurve_shape.py:192,
// physics_based.py:308) -- a boolean filter plus a sort per band per extractor.  Here the slice is
// read once (coalesced 8-byte lanes) and a single band-partitioned, time-sorted copy is built in
// LDS; the file-order arrays are kept because several features depend on row order
// (SURVEY.md §8a "order / tie traps").
//
// Sort rule: stable by (time, file index) -- pandas' default quicksort leaves ties undefined.
#pragma once
#include "wave.hpp"

namespace lcfe {

template <int CAP>
struct ObjLds {
    // file order
    double t[CAP], f[CAP], e[CAP];
    // band-partitioned (u,g,r,i,z,y segments), each segment sorted by (t, file index)
    double bt[CAP], bf[CAP], be[CAP];
    unsigned short bidx[CAP];   // file index of each band-sorted element
    unsigned char b[CAP];       // file order band codes (0..5, 255 unknown)
    int boff[8];                // boff[k]..boff[k+1] = segment of band k; boff[6] = #known-band points
    int n;                      // number of points
    int sorted;                 // 1 if file order is already non-decreasing in time
};

// Fill `L` from the global slice.  All lanes of the wave must call it.
template <class W, int CAP>
LCFE_FN void stage_object(const ObjIn& in, ObjLds<CAP>& L) {
    const int lane = W::lane();
    const int n = in.n;
    int cnt[6] = {0, 0, 0, 0, 0, 0};      // wave-uniform: ballots counted on the scalar unit
    bool ok = true;
    for (int base = 0; base < n; base += W::LANES) {
        const int i = base + lane;
        int bb = 256;
        if (i < n) {
            const double ti = in.t[i];
            L.t[i] = ti;
            L.f[i] = in.f[i];
            L.e[i] = in.e[i];
            const unsigned char b8 = in.b[i];
            L.b[i] = b8;
            bb = b8;
            if (i + 1 < n) ok = ok && (ti <= in.t[i + 1]);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) cnt[k] += popcll(W::ballot(bb == k));
    }
    const bool sorted = W::all(ok);
    int off = 0;
    if (lane == 0) L.boff[0] = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        off += cnt[k];
        if (lane == 0) L.boff[k + 1] = off;
    }
    if (lane == 0) { L.boff[7] = off; L.n = n; L.sorted = sorted ? 1 : 0; }
    W::sync();
    if (sorted) {
        // stable partition: position = segment start + number of earlier rows of the same band
        int run[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) run[k] = L.boff[k];
        for (int base = 0; base < n; base += W::LANES) {
            const int i = base + lane;
            const int bb = (i < n) ? (int)L.b[i] : 255;
            int pos = -1;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                unsigned long long m = W::ballot(bb == k);
                if (bb == k) pos = run[k] + W::prefix(m);
                run[k] += popcll(m);
            }
            if (pos >= 0) {
                L.bt[pos] = L.t[i];
                L.bf[pos] = L.f[i];
                L.be[pos] = L.e[i];
                L.bidx[pos] = (unsigned short)i;
            }
        }
    } else {
        // general path: rank of (t, i) among the rows of the same band
        for (int i = lane; i < n; i += W::LANES) {
            const int bb = L.b[i];
            if (bb >= 6) continue;
            const double ti = L.t[i];
            int r = 0;
            for (int j = 0; j < n; ++j) {
                const double tj = L.t[j];
                r += (L.b[j] == bb) && ((tj < ti) || (tj == ti && j < i));
            }
            const int pos = L.boff[bb] + r;
            L.bt[pos] = ti;
            L.bf[pos] = L.f[i];
            L.be[pos] = L.e[i];
            L.bidx[pos] = (unsigned short)i;
        }
    }
    W::sync();
}

// Order statistics by rank counting.  `x` = m wave-shared values, `keys` = m words of wave-shared
// scratch.  For every requested rank r (0-based position in numpy's sort order, NaN last) the value
// whose [count_less, count_less_or_equal) interval contains r is written to sel[t] -- equal values
// share an interval, so ties need no index tie-break.  The scan is register-blocked (4 own
// elements per lane against 4 keys per trip) so that LDS reads are pipelined instead of exposing
// one round trip per comparison.
template <class W, int NT>
LCFE_FN void wave_select_ranks(const double* x, int m, unsigned long long* keys, const int* ranks, double* sel) {
    const int lane = W::lane();
    for (int i = lane; i < m; i += W::LANES) keys[i] = sort_key(x[i]);
    W::sync();
    const int m4 = m & ~3;
    for (int base = 0; base < m; base += 4 * W::LANES) {
        unsigned long long ki[4];
        int clt[4] = {0, 0, 0, 0}, cle[4] = {0, 0, 0, 0};
        bool own[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = base + c * W::LANES + lane;
            own[c] = i < m;
            ki[c] = own[c] ? keys[i] : 0ull;
        }
        for (int j = 0; j < m4; j += 4) {
            const unsigned long long k0 = keys[j], k1 = keys[j + 1], k2 = keys[j + 2], k3 = keys[j + 3];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                clt[c] += (k0 < ki[c]) + (k1 < ki[c]) + (k2 < ki[c]) + (k3 < ki[c]);
                cle[c] += (k0 <= ki[c]) + (k1 <= ki[c]) + (k2 <= ki[c]) + (k3 <= ki[c]);
            }
        }
        for (int j = m4; j < m; ++j) {
            const unsigned long long kj = keys[j];
#pragma unroll
            for (int c = 0; c < 4; ++c) { clt[c] += (kj < ki[c]); cle[c] += (kj <= ki[c]); }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (!own[c]) continue;
            const int i = base + c * W::LANES + lane;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (clt[c] <= ranks[t] && ranks[t] < cle[c]) sel[t] = x[i];
        }
    }
    W::sync();
}

// ---- ascending sorting network over the LANES x KPL values a lane group holds in registers
// (element index = lane * KPL + r).  Bitonic merges in the direction-free form: the first step of a
// size-K merge compares i with its mirror i ^ (K-1), the following half-cleaners compare i with
// i ^ J for J = K/4 ... 1; the lower index always keeps the minimum.  Partners closer than KPL are
// other registers of the same lane, the rest are fetched from lane ^ (J / KPL) (DPP / swizzle).
// The values must not contain NaN (callers branch on that before; pad with +inf).
template <class W, int KPL, int K>
LCFE_FN void sort_flip_step(double (&v)[KPL]) {
    if constexpr (K <= KPL) {
#pragma unroll
        for (int r = 0; r < KPL; ++r) {
            const int q = r ^ (K - 1);
            if (r < q) { const double a = v[r], b = v[q]; v[r] = dmin(a, b); v[q] = dmax(a, b); }
        }
    } else {
        constexpr int ML = K / KPL - 1;
        const bool keep_min = (W::lane() & ((ML + 1) >> 1)) == 0;
        double p[KPL];
#pragma unroll
        for (int r = 0; r < KPL; ++r) p[r] = W::template xfetch<ML>(v[KPL - 1 - r]);
        // one compare + select instead of min, max and a select: the partner replaces the own value where
        // "partner < own" agrees with "this lane keeps the minimum" (NaN-free data; equal values may swap freely)
#pragma unroll
        for (int r = 0; r < KPL; ++r) v[r] = ((p[r] < v[r]) == keep_min) ? p[r] : v[r];
    }
}
template <class W, int KPL, int J>
LCFE_FN void sort_half_steps(double (&v)[KPL]) {
    if constexpr (J >= 1) {
        if constexpr (J < KPL) {
#pragma unroll
            for (int r = 0; r < KPL; ++r)
                if ((r & J) == 0) { const double a = v[r], b = v[r | J]; v[r] = dmin(a, b); v[r | J] = dmax(a, b); }
        } else {
            constexpr int ML = J / KPL;
            const bool keep_min = (W::lane() & ML) == 0;
            double p[KPL];
#pragma unroll
            for (int r = 0; r < KPL; ++r) p[r] = W::template xfetch<ML>(v[r]);
#pragma unroll
            for (int r = 0; r < KPL; ++r) v[r] = ((p[r] < v[r]) == keep_min) ? p[r] : v[r];
        }
        sort_half_steps<W, KPL, J / 2>(v);
    }
}
template <class W, int KPL, int K>
LCFE_FN void sort_merges(double (&v)[KPL]) {
    if constexpr (K <= W::LANES * KPL) {
        sort_flip_step<W, KPL, K>(v);
        sort_half_steps<W, KPL, K / 4>(v);
        sort_merges<W, KPL, K * 2>(v);
    }
}

// sorted[0..m) = x[0..m) in ascending order (x NaN-free, m <= LANES * KPL; `sorted` may not alias x).
// With one lane (host build) this is a plain insertion sort.
template <class W, int KPL>
LCFE_FN void group_sort_values(const double* x, int m, double* sorted) {
    if constexpr (W::LANES == 1) {
        for (int i = 0; i < m; ++i) {
            const double v = x[i];
            int j = i;
            while (j > 0 && sorted[j - 1] > v) { sorted[j] = sorted[j - 1]; --j; }
            sorted[j] = v;
        }
    } else {
        const int base = W::lane() * KPL;
        double v[KPL];
#pragma unroll
        for (int r = 0; r < KPL; ++r) v[r] = (base + r < m) ? x[base + r] : __builtin_inf();
        sort_merges<W, KPL, 2>(v);
#pragma unroll
        for (int r = 0; r < KPL; ++r)
            if (base + r < m) sorted[base + r] = v[r];
    }
    W::sync();
}

// median(|s_i - med|) of an ascending array s[0..m) whose median is `med` (finite): the deviations of
// the upper half [h, m) ascend, those of the lower half descend, so the two middle ranks of their
// merge follow from "how many of the r+1 smallest come from the upper half" = the number of
// candidates i with upper[i] < lower[r - i] (a monotone predicate, all candidates tested at once,
// one LDS round trip).  Every deviation is the same floating-point subtraction numpy performs, so
// the result is exact.  Uniform over the group.
template <class W>
LCFE_FN double mad_of_sorted(const double* s, int m, double med) {
    const int lane = W::lane();
    const int h = m / 2, a = m - h, b = h;
    const int r = (m - 1) / 2;                 // lower middle rank; the upper one is m / 2
    const int lo = (r + 1 - b > 0) ? r + 1 - b : 0, hi = (a < r + 1) ? a : r + 1;
    int cnt = 0;
    for (int base = lo; base < hi; base += W::LANES) {
        const int c = base + lane;
        const bool in = c < hi;
        const int ci = in ? c : lo;            // lo < hi here, so index lo is a valid candidate
        const bool p = in && (s[h + ci] - med < med - s[h - (r + 1 - ci)]);
        cnt += popcll(W::ballot(p));
    }
    const int i = lo + cnt, j = r + 1 - i;
    const double ua = s[h + ((i > 0) ? i - 1 : 0)] - med, la = med - s[(j > 0) ? h - j : 0];
    const double ub = s[h + ((i < a) ? i : 0)] - med, lb = med - s[(j < b) ? h - 1 - j : 0];
    double v_lo = (i > 0) ? ua : -__builtin_inf();
    v_lo = (j > 0 && la > v_lo) ? la : v_lo;
    if (r == m / 2) return v_lo;
    double v_hi = (i < a) ? ub : __builtin_inf();
    v_hi = (j < b && lb < v_hi) ? lb : v_hi;
    return (v_lo + v_hi) / 2.0;
}

// numpy.argmax semantics on a (wave-shared) array: index of the FIRST maximum; a NaN counts as
// the maximum (numpy propagates the first NaN).  Returns -1 for m == 0.  Uniform result.
template <class W>
LCFE_FN int wave_argmax_first(const double* x, int m) {
    const int lane = W::lane();
    double best = 0.0;
    int bi = 0x7fffffff;
    bool have = false;
    for (int i = lane; i < m; i += W::LANES) {
        const double v = x[i];
        if (!have) { best = v; bi = i; have = true; }
        else if (!is_nan(best) && (is_nan(v) || v > best)) { best = v; bi = i; }
    }
    // reduce (value desc, NaN first, index asc)
    // step 1: is there a NaN anywhere?  then answer = min index of a NaN
    const bool has_nan = W::any(have && is_nan(best));
    if (has_nan) {
        int c = (have && is_nan(best)) ? bi : 0x7fffffff;
        return W::min(c);
    }
    if (m <= 0) return -1;
    const double mx = W::max(have ? best : -__builtin_inf());
    int c = (have && best == mx) ? bi : 0x7fffffff;
    return W::min(c);
}

}  // namespace lcfe
