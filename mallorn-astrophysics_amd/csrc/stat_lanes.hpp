// stat_lanes.hpp -- statistics with EIGHT light curves per wavefront and one lane per band.
//
// The one-object-per-wavefront kernels (stat_lean.hpp) spend most of their instructions on things that
// are not statistics: cross-lane partner fetches of the sorting networks, 15 DPP reductions per group,
// padding of 23-row bands to 32 slots in 8 of 64 lanes.  Here a light curve owns an 8-lane group and a
// BAND OWNS A LANE: the band's fluxes sit in that lane's registers, the moment passes are plain serial
// loops without any reduction, the band sort is a sorting network on registers (v_min/v_max pairs, no
// partner fetch) and eight light curves share every instruction.  r and i, the two long bands of the
// survey's cadence, are split in time over two lanes each, so that a register array of CAP values
// holds bands of CAP rows (u, g, z, y) and 2 CAP rows (r, i):
//
//     lane of the group   0   1   2        3         4   5   6        7
//     rows                u   g   r first  r second  z   y   i first  i second
//
// One workgroup (= one wavefront) takes one batch of eight light curves; lane j of a group first holds rows j, j + 8,
// ... of its light curve (t, f, e, band code) in registers from ONE round of loads.  Phases (one LDS buffer of
// 64 columns x (CAP + 1) doubles per wavefront, reused):
//   A  positions: a row's slot = number of earlier rows of its band (ballots over the 8 rows of a trip); per row also
//      the all-rows slope and the "rows ascend in time" check from the file-order neighbours (next lane / next trip)
//      and the SNR term |f| / e
//   F  fluxes -> LDS (column = lane that owns the band half) -> registers v[0..CAP); sums, NaN flags
//   T  times  -> LDS -> band slopes against the register-resident fluxes
//   Q  SNR terms -> LDS -> sums
//   then the two centred passes (band and all-rows moments side by side, select-free on a copy padded with the band
//   mean; the all-rows sums are the 8-lane sums of the lanes' partials), the register sort, and the all-rows order
//   statistics from a bitonic MERGE of the eight sorted lanes (whose first level is the merge of the r and i halves).
//   Extrema and order statistics are read off LDS dumps of the sorted registers by one lane per sequence; the
//   all-rows columns are parked in LDS as they get ready (the kernel has no register to spare: 256 per lane, no spill).
// Which light curves come here is decided by stat_plan_kernel (lcfe.hip) from the rows per band; a light curve whose
// rows turn out not to ascend in time is appended to the general kernel's list.  Same arithmetic per element as
// stat.hpp (two-pass moments, exact counts, numpy's percentile interpolation, exact MAD); the sums are associated
// differently, and the slope quotients go through the reciprocal (stat_slope).
#pragma once
#include <utility>
#include "stat.hpp"

namespace lcfe {

// Batcher's odd-even merge sort for N = 2^k keys as a compile-time list of compare-exchanges
// (63 / 191 / 543 for 16 / 32 / 64 keys; the bitonic network needs 80 / 240 / 672).
template <int N>
struct OddEvenNet {
    int count;
    unsigned char a[N * 10], b[N * 10];
    constexpr OddEvenNet() : count(0), a{}, b{} {
        for (int p = 1; p < N; p *= 2)
            for (int k = p; k >= 1; k /= 2)
                for (int j = k % p; j <= N - 1 - k; j += 2 * k)
                    for (int i = 0; i <= ((k - 1 < N - j - k - 1) ? k - 1 : N - j - k - 1); ++i)
                        if ((i + j) / (p * 2) == (i + j + k) / (p * 2)) {
                            a[count] = (unsigned char)(i + j);
                            b[count] = (unsigned char)(i + j + k);
                            ++count;
                        }
    }
};
template <int N>
inline constexpr OddEvenNet<N> kOddEvenNet{};

}  // namespace lcfe

#if defined(__HIPCC__)
namespace lcfe {

template <int N, int A, int B>
__device__ __forceinline__ void reg_exchange(double (&w)[N]) {
    const double lo = dmin(w[A], w[B]), hi = dmax(w[A], w[B]);
    w[A] = lo;
    w[B] = hi;
}
template <int N, size_t... I>
__device__ __forceinline__ void reg_sort_seq(double (&w)[N], std::index_sequence<I...>) {
    (reg_exchange<N, kOddEvenNet<N>.a[I], kOddEvenNet<N>.b[I]>(w), ...);
}
// ascending sort of N register-resident keys (no NaN among them)
template <int N>
__device__ __forceinline__ void reg_sort(double (&w)[N]) {
    reg_sort_seq<N>(w, std::make_index_sequence<kOddEvenNet<N>.count>{});
}

// half-cleaners of a bitonic sequence held in the registers of one lane: distances D, D/2, .., 1
template <int N, int D>
__device__ __forceinline__ void reg_half_cleaners(double (&w)[N]) {
    if constexpr (D >= 1) {
#pragma unroll
        for (int r = 0; r < N; ++r)
            if ((r & D) == 0) {
                const double lo = dmin(w[r], w[r + D]), hi = dmax(w[r], w[r + D]);
                w[r] = lo;
                w[r + D] = hi;
            }
        reg_half_cleaners<N, D / 2>(w);
    }
}
// half-cleaner between lanes l and l ^ X (same register): the lane with the bit clear keeps the minimum
template <int N, int X>
__device__ __forceinline__ void lane_half_cleaner(double (&w)[N], bool keep_min) {
#pragma unroll
    for (int r = 0; r < N; ++r) {
        const double p = lane_xor_fetch<X>(w[r]);
        w[r] = ((p < w[r]) == keep_min) ? p : w[r];
    }
}
// Merge step over blocks of 2 D lanes: the sequences (lane-major: index = lane * N + register) of the lower and
// the upper D lanes are ascending; afterwards the 2 D lanes hold their union, ascending.  First the element-reversed
// compare (partner lane l ^ (2 D - 1), register N - 1 - r), then half-cleaners at lane distances D / 2 .. 1 and
// register distances N / 2 .. 1.
template <int N, int D>
__device__ __forceinline__ void lane_merge(double (&w)[N], int lane_in_group) {
    {
        const bool keep_min = (lane_in_group & D) == 0;
#pragma unroll
        for (int r = 0; r < N / 2; ++r) {
            const double p1 = lane_xor_fetch<2 * D - 1>(w[N - 1 - r]), p2 = lane_xor_fetch<2 * D - 1>(w[r]);
            w[r] = ((p1 < w[r]) == keep_min) ? p1 : w[r];
            w[N - 1 - r] = ((p2 < w[N - 1 - r]) == keep_min) ? p2 : w[N - 1 - r];
        }
    }
    if constexpr (D >= 8) lane_half_cleaner<N, 4>(w, (lane_in_group & 4) == 0);
    if constexpr (D >= 4) lane_half_cleaner<N, 2>(w, (lane_in_group & 2) == 0);
    if constexpr (D >= 2) lane_half_cleaner<N, 1>(w, (lane_in_group & 1) == 0);
    reg_half_cleaners<N, N / 2>(w);
}

// The lanes kernels index one LDS buffer with computed slots (band position tables, merge dumps, the output rows).
// Release builds use the plain pointer.  A -DLCFE_DEBUG build (make debug; SURVEY.md section 5 "sanitizer" row: the GPU
// AddressSanitizer is not available on this pool) replaces it by a bounds-checked view: an index outside the buffer is
// counted in g_lanes_check[0] (the access is redirected to slot 0 instead of faulting), and the kernel frames the buffer
// with canary words it verifies before it ends (g_lanes_check[1]).  lcfe_debug_lanes_check() reads the counters.
#ifdef LCFE_DEBUG
__device__ unsigned int g_lanes_check[4];      // out-of-range indices, damaged canaries, last bad index, its buffer length
// (branch-free: a bad index is clamped to slot 0 and counted in a per-lane register the kernel flushes when it ends --
//  hundreds of inlined divergent branches around the accesses changed the code the compiler produced for the rest)
struct LanesBuf {
    double* p;
    int n;
    unsigned int* bad;      // per-lane counter (a register of the kernel), [1] = last bad index
    __device__ __forceinline__ double& operator[](int i) const {
        const bool oob = (unsigned int)i >= (unsigned int)n;
        bad[0] += oob ? 1u : 0u;
        bad[1] = oob ? (unsigned int)i : bad[1];
        return p[oob ? 0 : i];
    }
    __device__ __forceinline__ LanesBuf operator+(int k) const { return LanesBuf{p + k, n - k, bad}; }
};
__device__ __forceinline__ LanesBuf lanes_buf(double* p, int n, unsigned int* bad) { return LanesBuf{p, n, bad}; }
// plain pointer to `count` doubles of the view, for the helpers that take one (the range is checked once)
__device__ __forceinline__ double* lanes_raw(const LanesBuf& b, int count) {
    const bool oob = count > b.n || b.n < 0;
    b.bad[0] += oob ? 1u : 0u;
    b.bad[1] = oob ? (unsigned int)count : b.bad[1];
    return b.p;
}
#else
using LanesBuf = double*;
__device__ __forceinline__ LanesBuf lanes_buf(double* p, int, unsigned int*) { return p; }
__device__ __forceinline__ double* lanes_raw(LanesBuf b, int) { return b; }
#endif

template <int CAP>
struct StatLanesLds {
    static constexpr int STRIDE = CAP + 1;       // odd number of doubles: a column per lane without bank conflicts
    double buf[64 * STRIDE];
    double all_rows[8 * 17];                     // the all-rows columns of the eight light curves, stored as they get ready
};

// element `idx` of an ascending sequence dumped lane-major from column `base` on (CAP registers per lane)
template <int CAP>
__device__ __forceinline__ double lanes_seq(LanesBuf buf, int base, int idx) {
    constexpr int SH = (CAP == 16) ? 4 : ((CAP == 32) ? 5 : 6);
    return buf[base + idx + (idx >> SH)];
}

// median, inter-quartile range and MAD of the ascending NaN-free sequence of m >= 1 values at `base`
template <int CAP>
__device__ __forceinline__ void lanes_order_stats(LanesBuf buf, int base, int m, double& med, double& iqr, double& mad) {
    auto S = [&](int i) { return lanes_seq<CAP>(buf, base, i); };
    const int r_lo = (m - 1) / 2, r_hi = m / 2;
    const double v25 = 0.25 * (m - 1), v75 = 0.75 * (m - 1);
    const int r25 = (int)floor(v25), r75 = (int)floor(v75);
    const int r25h = (r25 + 1 < m) ? r25 + 1 : m - 1, r75h = (r75 + 1 < m) ? r75 + 1 : m - 1;
    const double a0 = S(r_lo), a1 = S(r_hi), q0 = S(r25), q1 = S(r25h), q2 = S(r75), q3 = S(r75h);
    med = (r_lo == r_hi) ? a0 : (a0 + a1) / 2.0;
    iqr = (m > 1) ? np_lerp(q2, q3, v75 - r75) - np_lerp(q0, q1, v25 - r25) : 0.0;
    if (!(med - med == 0.0)) { mad = qnan(); return; }
    // stage.hpp::mad_of_sorted with the monotone predicate bisected by this lane alone
    const int h = m / 2, a = m - h, b = h, r = r_lo;
    const int lo = (r + 1 - b > 0) ? r + 1 - b : 0, hi = (a < r + 1) ? a : r + 1;
    int L = lo, H = hi;
    while (L < H) {
        const int c = (L + H) >> 1;
        const bool p = S(h + c) - med < med - S(h - (r + 1 - c));
        L = p ? c + 1 : L;
        H = p ? H : c;
    }
    const int i = L, j = r + 1 - i;
    const double ua = S(h + ((i > 0) ? i - 1 : 0)) - med, la = med - S((j > 0) ? h - j : 0);
    const double ub = S(h + ((i < a) ? i : 0)) - med, lb = med - S((j < b) ? h - 1 - j : 0);
    double v_lo = (i > 0) ? ua : -__builtin_inf();
    v_lo = (j > 0 && la > v_lo) ? la : v_lo;
    double v_hi = (i < a) ? ub : __builtin_inf();
    v_hi = (j < b && lb < v_hi) ? lb : v_hi;
    mad = (r == m / 2) ? v_lo : (v_lo + v_hi) / 2.0;
}

// x of the partner lane of a split band (lanes 2|3 and 6|7 of a group)
__device__ __forceinline__ double lanes_pair(double x) { return lane_xor_fetch<1>(x); }
__device__ __forceinline__ int lanes_pair(int x) { return lane_xor_fetch<1>(x); }

// Eight light curves (one per 8-lane group: `obj` < 0 = none; CSR rows [s1, e1)) -> their 123 columns, or list
// `fallback_list`.  ITERS = rows / 8 a light curve of the list may have (band code 256 = no row -- 255 is a code a file
// may hold).  The rows are loaded in predicated blocks of four trips: one basic block of unconditional loads, a
// persistent batch loop around this function, or exec-masked element loops all made the compiler spill hundreds of
// registers (profiles/r02_stat_instruction_budget.md).
template <int CAP, int ITERS>
__device__ __forceinline__ void stat_lanes_batch(const double* gt, const double* gf, const double* ge, const uint8_t* gb, int obj,
                                              int64_t s1, int64_t e1, LanesBuf buf, LanesBuf all_rows, double* out, int ld, int col0,
                                              int* fallback_list, int* fallback_count) {
    using G = GroupDev<8>;
    constexpr int STRIDE = StatLanesLds<CAP>::STRIDE;
    constexpr int BLK = 4;
    static_assert(ITERS % BLK == 0 && ITERS <= CAP, "rows per lane");
    const int lane = threadIdx.x & 63, j = lane & 7, g8 = lane & 56;
    const int n = (int)(e1 - s1);
    const bool has_obj = obj >= 0;
    bool fit = has_obj && n >= 1 && n <= 8 * ITERS;
    const int nr = fit ? n : 0;
    int nmax = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int v_ = __builtin_amdgcn_readlane(nr, 8 * k); nmax = (v_ > nmax) ? v_ : nmax; }
    const int iters = (nmax + 7) >> 3;
    // ---- rows -> registers
    double rt[ITERS], rf[ITERS], rq[ITERS];
    int code[ITERS];
    {
        const double *pt = gt + s1, *pf = gf + s1, *pe = ge + s1;
        const uint8_t* pb = gb + s1;
#pragma unroll
        for (int i0 = 0; i0 < ITERS; i0 += BLK) {
            if (i0 < iters) {
#pragma unroll
                for (int q = 0; q < BLK; ++q) {
                    const int row = (i0 + q) * 8 + j;
                    const bool ok = row < nr;
                    code[i0 + q] = ok ? (int)pb[row] : 256;
                    rt[i0 + q] = ok ? pt[row] : 0.0;
                    rf[i0 + q] = ok ? pf[row] : 0.0;
                    rq[i0 + q] = ok ? pe[row] : 0.0;
                }
            }
        }
    }

    // ---- A: slot of every row inside its band = rows of that band before it (ballots over the 8 rows of a trip)
    int cnt[6] = {0, 0, 0, 0, 0, 0};
    bool known = true;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        if ((it & ~(BLK - 1)) < iters) {
            const int b = code[it];
            known = known && (b < 6 || b == 256);
            unsigned int mine = 0;
            int c0 = 0;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const unsigned int m8 = (unsigned int)(__ballot(b == k) >> g8) & 0xFFu;
                if (b == k) { mine = m8; c0 = cnt[k]; }
                cnt[k] += __builtin_popcount(m8);
            }
            const int pos = c0 + __builtin_popcount(mine & ((1u << j) - 1u));
            code[it] = ((b < 6) ? b : 7) | (pos << 8);
        }
    }
    fit = fit && G::all(known) && cnt[0] <= CAP && cnt[1] <= CAP && cnt[4] <= CAP && cnt[5] <= CAP && cnt[2] <= 2 * CAP &&
          cnt[3] <= 2 * CAP;
    const int N = fit ? n : 0;
    const int h1r = (cnt[2] + 1) >> 1, h1i = (cnt[3] + 1) >> 1;
    // this lane's share: band, rows
    const int band = (j == 0) ? 0 : (j == 1) ? 1 : (j <= 3) ? 2 : (j == 4) ? 4 : (j == 5) ? 5 : 3;
    const bool split = (j & 2) != 0, first = split && (j & 1) == 0;
    int mband = (band == 0) ? cnt[0] : (band == 1) ? cnt[1] : (band == 2) ? cnt[2] : (band == 3) ? cnt[3] : (band == 4) ? cnt[4] : cnt[5];
    const int h1 = (band == 2) ? h1r : h1i;
    int m = !split ? mband : (first ? h1 : mband - h1);
    if (!fit) { m = 0; mband = 0; }
    const bool bridge = first && mband > m;                 // the pair (last row of the first half, first row of the second)
    const int col = lane * STRIDE;

    // ---- per row: LDS slot (-1: none); the all-rows slope and the order check from the file neighbours (row + 1 = next
    //      lane, or lane 0 of the next trip); the SNR term in place of the error (-0.0 = error bar not positive: the
    //      reference leaves the row out; a term is never negative)
    bool ordered = true, a_snan = false;
    double a_slope = -1.0;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        if ((it & ~(BLK - 1)) < iters) {
            const int b = code[it] & 0xFF, pos = code[it] >> 8;
            const bool second = (b == 2 && pos >= h1r) || (b == 3 && pos >= h1i);
            const int hb = (b == 2) ? h1r : h1i;
            const int dl = (b == 0) ? 0 : (b == 1) ? 1 : (b == 2) ? 2 : (b == 3) ? 6 : (b == 4) ? 4 : 5;
            code[it] = (b < 6 && fit) ? (g8 + dl + (second ? 1 : 0)) * STRIDE + (second ? pos - hb : pos) : -1;
            const int row = it * 8 + j;
            const bool has = row + 1 < N;
            const double tn_l = G::template dpp<0x101>(rt[it]), fn_l = G::template dpp<0x101>(rf[it]);   // row_shl:1
            const double tn_w = G::template dpp<0x117>(rt[(it + 1 < ITERS) ? it + 1 : it]),              // row_shr:7
                         fn_w = G::template dpp<0x117>(rf[(it + 1 < ITERS) ? it + 1 : it]);
            const double t1 = (j == 7) ? tn_w : tn_l, f1 = (j == 7) ? fn_w : fn_l;
            ordered = ordered && !(has && !(rt[it] <= t1));
            const double dt = t1 - rt[it];
            const double sl = stat_slope(f1 - rf[it], dt);
            const bool valid = has && dt > 0;
            a_snan = a_snan || (valid && is_nan(sl));
            a_slope = (valid && sl > a_slope) ? sl : a_slope;
            rq[it] = (rq[it] > 0) ? fabs(rf[it]) / rq[it] : -0.0;
        }
    }

    // ---- F: fluxes -> lanes (the columns are cleared first: the slots behind a lane's rows read as 0.0 in every phase)
#pragma unroll
    for (int i = 0; i < STRIDE; ++i) buf[col + i] = 0.0;
    G::sync();
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
        if ((it & ~(BLK - 1)) < iters && code[it] >= 0) buf[code[it]] = rf[it];
    G::sync();
    double v[CAP];
#pragma unroll
    for (int i = 0; i < CAP; ++i) v[i] = buf[col + i];
    const double f_last = buf[col + ((m > 0) ? m - 1 : 0)], f_next = buf[col + STRIDE - ((j == 7) ? STRIDE : 0)];
    double s;
    bool nanf = false;
    {
        double sa[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < CAP; ++i) {
            sa[i & 3] += v[i];
            nanf = nanf || is_nan(v[i]);
        }
        s = (sa[0] + sa[1]) + (sa[2] + sa[3]);
    }
    G::sync();

    // ---- T: times -> lanes, band slopes against the register-resident fluxes
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
        if ((it & ~(BLK - 1)) < iters && code[it] >= 0) buf[code[it]] = rt[it];
    G::sync();
    double slope = -1.0, tmn, tmx;
    bool snan = false;
    {
        double tc = buf[col];
        tmn = tc;
#pragma unroll
        for (int i = 0; i + 1 < CAP; ++i) {
            const double tn = buf[col + i + 1];
            const double dt = tn - tc;
            const double sl = stat_slope(v[i + 1] - v[i], dt);
            const bool valid = i + 1 < m && dt > 0;
            snan = snan || (valid && is_nan(sl));
            slope = (valid && sl > slope) ? sl : slope;
            tc = tn;
        }
        tmx = buf[col + ((m > 0) ? m - 1 : 0)];
        if (bridge) {
            const double dt = buf[col + STRIDE] - tmx;
            const double sl = stat_slope(f_next - f_last, dt);
            const bool valid = dt > 0;
            snan = snan || (valid && is_nan(sl));
            slope = (valid && sl > slope) ? sl : slope;
        }
    }
    G::sync();

    // ---- Q: SNR terms -> lanes
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
        if ((it & ~(BLK - 1)) < iters && code[it] >= 0) buf[code[it]] = rq[it];
    G::sync();
    double snr;
    int nsnr = m;
    {
        double qa[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < CAP; ++i) {
            const double qv = buf[col + i];
            qa[i & 3] += qv;
            nsnr -= (int)((unsigned long long)__builtin_bit_cast(long long, qv) >> 63);
        }
        snr = (qa[0] + qa[1]) + (qa[2] + qa[3]);
    }
    G::sync();

    // ---- pass-1 totals: per band (the two halves of r and i combined) and over the light curve
    const double sA = G::sum(s), snrA = G::sum(snr);
    const int nsnrA = G::sum(nsnr);
    const bool nanA = G::any(nanf);
    const double a_slopeA = G::max(a_slope);
    const bool a_snanA = G::any(a_snan);
    const bool orderedA = G::all(ordered);
    const double tmnA = G::min((m > 0) ? tmn : __builtin_inf()), tmxA = G::max((m > 0) ? tmx : -__builtin_inf());
    if (split) {
        const int p_nan = lanes_pair(nanf ? 1 : 0), p_snan = lanes_pair(snan ? 1 : 0);   // fetched by every lane (no short-circuit)
        s += lanes_pair(s);
        snr += lanes_pair(snr);
        nsnr += lanes_pair(nsnr);
        nanf = nanf | (p_nan != 0);
        const double p_tmn = lanes_pair(tmn), p_tmx = lanes_pair(tmx), p_slope = lanes_pair(slope);
        const int p_m = mband - m;
        tmn = first ? tmn : p_tmn;                     // a band with rows has rows in its first half
        tmx = first ? ((p_m > 0) ? p_tmx : tmx) : ((m > 0) ? tmx : p_tmx);
        slope = (p_slope > slope) ? p_slope : slope;
        snan = snan | (p_snan != 0);
    }
    const double mean = s / mband, meanA = sA / N;
    LanesBuf oa = all_rows + (g8 >> 3) * 17;              // the all-rows columns leave the registers as soon as they are known
    if (j == 3) {
        oa[0] = (double)N;
        oa[1] = meanA;
        oa[13] = (N > 1) ? (a_snanA ? qnan() : (a_slopeA < 0 ? 0.0 : a_slopeA)) : 0.0;
        oa[14] = (nsnrA > 0) ? snrA / nsnrA : qnan();
        oa[15] = (N > 1) ? (tmxA - tmnA) : 0.0;
        oa[16] = (N > 1) ? (tmxA - tmnA) / (double)(N - 1) : 0.0;
    }

    // ---- behind the lane's rows the band mean: the centred band passes need no per-element mask (a padded slot adds
    //      exactly 0 and counts as "inside"); the all-rows terms are masked
#pragma unroll
    for (int i = 0; i < CAP; ++i) v[i] = (i < m) ? v[i] : mean;
    // ---- pass 2
    double m2, m2A;
    {
        double a[2] = {0.0, 0.0}, aA[2] = {0.0, 0.0};
#pragma unroll
        for (int i = 0; i < CAP; ++i) {
            const double d = v[i] - mean, dA = (i < m) ? v[i] - meanA : 0.0;
            a[i & 1] += d * d;
            aA[i & 1] += dA * dA;
        }
        m2 = a[0] + a[1];
        m2A = aA[0] + aA[1];
    }
    m2A = G::sum(m2A);
    if (split) m2 += lanes_pair(m2);
    const double sd = (mband > 1) ? sqrt(m2 / mband) : 0.0, sdA = (N > 1) ? sqrt(m2A / N) : 0.0;

    // ---- pass 3 (the lanes of a light curve whose spread is not positive carry garbage that is dropped below)
    double s3, s4, s3A, s4A;
    int c1 = 0, c2 = 0, c1A = 0, c2A = 0;
    {
        const double inv = 1.0 / sd, sd2 = 2.0 * sd, invA = 1.0 / sdA, sdA2 = 2.0 * sdA;
        double a3[2] = {0.0, 0.0}, a4[2] = {0.0, 0.0}, b3[2] = {0.0, 0.0}, b4[2] = {0.0, 0.0};
#pragma unroll
        for (int i = 0; i < CAP; ++i) {
            const double d = v[i] - mean, dA = (i < m) ? v[i] - meanA : 0.0;
            const double zz = d * inv, zA = dA * invA;
            const double z2 = zz * zz, zA2 = zA * zA;
            a3[i & 1] += z2 * zz;
            a4[i & 1] += z2 * z2;
            b3[i & 1] += zA2 * zA;
            b4[i & 1] += zA2 * zA2;
            const double ad = fabs(d), adA = fabs(dA);
            c1 += (ad > sd) ? 1 : 0;
            c2 += (ad > sd2) ? 1 : 0;
            c1A += (adA > sdA) ? 1 : 0;
            c2A += (adA > sdA2) ? 1 : 0;
        }
        s3 = a3[0] + a3[1];
        s4 = a4[0] + a4[1];
        s3A = b3[0] + b3[1];
        s4A = b4[0] + b4[1];
    }
    s3A = G::sum(s3A);
    s4A = G::sum(s4A);
    c1A = G::sum(c1A);
    c2A = G::sum(c2A);
    if (split) {
        s3 += lanes_pair(s3);
        s4 += lanes_pair(s4);
        c1 += lanes_pair(c1);
        c2 += lanes_pair(c2);
    }
    auto finish = [](int cnt_, double sd_, double s3_, double s4_, int c1_, int c2_, double& skew, double& kurt, double& b1, double& b2) {
        skew = 0.0; kurt = 0.0; b1 = 0.0; b2 = 0.0;
        if (sd_ > 0) {
            if (cnt_ > 2) skew = s3_ / cnt_;
            if (cnt_ > 3) kurt = s4_ / cnt_ - 3.0;
            b1 = (double)c1_ / cnt_;
            b2 = (double)c2_ / cnt_;
        } else if (is_nan(sd_)) {
            skew = (cnt_ > 2) ? qnan() : 0.0;
            kurt = (cnt_ > 3) ? qnan() : 0.0;
        }
    };
    double skew, kurt, b1, b2, skewA, kurtA, b1A, b2A;
    finish(mband, sd, s3, s4, c1, c2, skew, kurt, b1, b2);
    finish(N, sdA, s3A, s4A, c1A, c2A, skewA, kurtA, b1A, b2A);
    if (j == 3) {
        oa[2] = sdA;
        oa[6] = skewA;
        oa[7] = kurtA;
        oa[11] = b1A;
        oa[12] = b2A;
    }


    // ---- order statistics and extrema: register sort per lane, then merges across the lanes
    double w[CAP];
#pragma unroll
    for (int i = 0; i < CAP; ++i) w[i] = (i < m) ? v[i] : __builtin_inf();
    reg_sort<CAP>(w);
    double mn = w[0], mx = -__builtin_inf();                // the maximum is read off the dumps (rank rows - 1)
    double med = qnan(), iqr = qnan(), mad = qnan();
#pragma unroll
    for (int i = 0; i < CAP; ++i) buf[col + i] = w[i];
    G::sync();
    if (m > 0) mx = buf[col + m - 1];
    if (!split && m > 0 && !nanf) lanes_order_stats<CAP>(buf, col, m, med, iqr, mad);
    G::sync();
    double mnA = G::min(mn), mxA = G::max(mx);
    if (split) {
        mn = dmin(mn, lanes_pair(mn));
        mx = dmax(mx, lanes_pair(mx));
    }
    lane_merge<CAP, 1>(w, j);
#pragma unroll
    for (int i = 0; i < CAP; ++i) buf[col + i] = w[i];
    G::sync();
    if (first && mband > 0 && !nanf) lanes_order_stats<CAP>(buf, col, mband, med, iqr, mad);
    G::sync();
    lane_merge<CAP, 2>(w, j);
    lane_merge<CAP, 4>(w, j);
#pragma unroll
    for (int i = 0; i < CAP; ++i) buf[col + i] = w[i];
    G::sync();
    double medA = qnan(), iqrA = qnan(), madA = qnan();
    if (j == 3 && N > 0 && !nanA) lanes_order_stats<CAP>(buf, g8 * STRIDE, N, medA, iqrA, madA);
    G::sync();
    if (mband <= 1) iqr = 0.0;                               // statistical.py:86: 0 unless the group has two rows
    if (N <= 1) iqrA = 0.0;
    if (nanf) { mn = qnan(); mx = qnan(); }
    if (nanA) { mnA = qnan(); mxA = qnan(); }
    if (j == 3) {
        oa[3] = mnA;
        oa[4] = mxA;
        oa[8] = mxA - mnA;
        oa[5] = medA;
        oa[9] = madA;
        oa[10] = iqrA;
    }

    // ---- the 123 columns of every light curve -> LDS rows -> global
    LanesBuf o = buf + (g8 >> 3) * 128;
    if (fit) {
        if (!split || first) {
            double* ob = lanes_raw(o + 17 * band, 17);
            if (mband == 0) stat_empty_group(ob, nullptr);
            else stat_write17(ob, mband, mean, sd, mn, mx, med, skew, kurt, mad, iqr, b1, b2, slope, snan, snr, nsnr, tmn, tmx);
        }
    }
    G::sync();
    if (fit) {
#pragma unroll
        for (int c = j; c < 17; c += 8) o[102 + c] = oa[c];
    }
    G::sync();
    if (fit && j == 0) stat_cross_band(lanes_raw(o, STAT_NCOL));
    G::sync();
    // rows to global memory, one light curve at a time on the whole wavefront (two 512-byte stores per row)
    const bool done = fit && orderedA;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int obj_r = __builtin_amdgcn_readlane(done ? obj : -1, 8 * r);
        if (obj_r >= 0) {
            double* row = out + (int64_t)obj_r * ld + col0;
            LanesBuf src = buf + r * 128;
            row[lane] = src[lane];
            if (lane + 64 < STAT_NCOL) row[lane + 64] = src[lane + 64];
        }
    }
    if (!done && has_obj && j == 0) {
        const int slot = atomicAdd(fallback_count, 1);
        fallback_list[slot] = obj;
    }
    G::sync();
}

// One workgroup = one batch (`batch`) of eight consecutive list entries; `buf`: 64 x (CAP + 1) doubles of LDS.
template <int CAP, int ITERS>
__device__ __forceinline__ void stat_lanes_run(const int64_t* offsets, const double* gt, const double* gf, const double* ge,
                                               const uint8_t* gb, const int* list, int count, int batch, LanesBuf buf, LanesBuf all_rows,
                                               double* out, int ld, int col0, int* fallback_list, int* fallback_count) {
    const int g = (threadIdx.x & 63) >> 3;
    const int64_t base = (int64_t)batch * 8;
    int obj = -1;
    int64_t s1 = 0, e1 = 0;
    if (base + g < count) {
        obj = list[base + g];
        s1 = offsets[obj];
        e1 = offsets[obj + 1];
    }
    stat_lanes_batch<CAP, ITERS>(gt, gf, ge, gb, obj, s1, e1, buf, all_rows, out, ld, col0, fallback_list, fallback_count);
}

}  // namespace lcfe
#endif
