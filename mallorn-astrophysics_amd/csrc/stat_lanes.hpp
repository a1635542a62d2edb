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
// Phases (one LDS buffer of 64 columns x (CAP + 1) doubles per wavefront, reused):
//   A  positions: 8 rows of every light curve per trip; a row's slot = number of earlier rows of its band
//      (ballots); nothing but the band byte is read
//   F  fluxes -> LDS (column = lane that owns the band half) -> registers v[0..CAP]; sums, extrema
//   T  times  -> LDS -> band slopes against the register-resident fluxes; the all-rows slope and the
//      "rows ascend in time" check are taken on the fly from the file-order neighbours
//   Q  |f| / e -> LDS -> SNR sums
//   then the two centred passes (band and all-rows moments side by side, the all-rows sums being the
//   8-lane sums of the lanes' partials), the register sort, and the all-rows order statistics from a
//   bitonic MERGE of the eight sorted lanes (the first level of which is the merge of the r and i halves).
//   Order statistics are read off LDS dumps of the sorted registers by one lane per sequence.
// A light curve that does not fit (a band longer than its lane(s), unknown band code, rows not in time
// order) goes to the general kernel's list.  Same arithmetic per element as stat.hpp (two-pass moments,
// exact counts, numpy's percentile interpolation, exact MAD); only the association of the sums differs.
#pragma once
#include <utility>
#include "stat.hpp"

#if defined(__HIPCC__)
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
    if constexpr (D >= 4) lane_half_cleaner<N, 2>(w, (lane_in_group & 2) == 0);
    if constexpr (D >= 2) lane_half_cleaner<N, 1>(w, (lane_in_group & 1) == 0);
    reg_half_cleaners<N, N / 2>(w);
}

template <int CAP>
struct StatLanesLds {
    static constexpr int STRIDE = CAP + 1;       // odd number of doubles: a column per lane without bank conflicts
    double buf[64 * STRIDE];
};

// element `idx` of an ascending sequence dumped lane-major from column `base` on (CAP registers per lane)
template <int CAP>
__device__ __forceinline__ double lanes_seq(const double* buf, int base, int idx) {
    constexpr int SH = (CAP == 16) ? 4 : ((CAP == 32) ? 5 : 6);
    return buf[base + idx + (idx >> SH)];
}

// median, inter-quartile range and MAD of the ascending NaN-free sequence of m >= 1 values at `base`
template <int CAP>
__device__ __forceinline__ void lanes_order_stats(const double* buf, int base, int m, double& med, double& iqr, double& mad) {
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

// Eight light curves (list entries list[0..nk)) -> their 123 columns, or the general kernel's list.
template <int CAP>
__device__ __forceinline__ void stat_lanes_batch(const int64_t* offsets, const double* gt, const double* gf, const double* ge,
                                                 const uint8_t* gb, const int* list, int nk, StatLanesLds<CAP>& L,
                                                 double* out, int ld, int col0, int* fallback_list, int* fallback_count) {
    using G = GroupDev<8>;
    constexpr int STRIDE = StatLanesLds<CAP>::STRIDE;
    constexpr int BLK = 4;                                  // rows of a light curve per lane between two waits on memory
    double* buf = L.buf;
    const int lane = threadIdx.x & 63, g = lane >> 3, j = lane & 7, g8 = g << 3;
    int obj = -1, n = 0;
    int64_t s0 = 0;
    if (g < nk) {
        obj = list[g];
        s0 = offsets[obj];
        n = (int)(offsets[obj + 1] - s0);
    }
    const double *pt = gt + s0, *pf = gf + s0, *pe = ge + s0;
    const uint8_t* pb = gb + s0;
    int nmax = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int nk_ = __builtin_amdgcn_readlane(n, 8 * k); nmax = (nk_ > nmax) ? nk_ : nmax; }
    bool fit = n >= 1 && n <= 8 * CAP;
    const int nr = fit ? n : 0;                             // rows this group stages
    nmax = (nmax < 8 * CAP) ? nmax : 8 * CAP;
    const int iters = (nmax + 7) >> 3;

    // ---- A: slot of every row inside its band
    int code[CAP];
    int cnt[6] = {0, 0, 0, 0, 0, 0};
    bool known = true;
#pragma unroll
    for (int i0 = 0; i0 < CAP; i0 += BLK) {
        if (i0 < iters) {
            int bb[BLK];
#pragma unroll
            for (int q = 0; q < BLK; ++q) {
                const int row = (i0 + q) * 8 + j;
                bb[q] = (row < nr) ? (int)pb[row] : 256;            // 256 = no row (255 is a band code a file may hold)
            }
#pragma unroll
            for (int q = 0; q < BLK; ++q) {
                const int b = bb[q];
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
                code[i0 + q] = ((b < 6) ? b : 7) | (pos << 8);
            }
        }
    }
    fit = fit && G::all(known) && cnt[0] <= CAP && cnt[1] <= CAP && cnt[4] <= CAP && cnt[5] <= CAP && cnt[2] <= 2 * CAP &&
          cnt[3] <= 2 * CAP;
    const int N = fit ? n : 0;
    const int h1r = (cnt[2] + 1) >> 1, h1i = (cnt[3] + 1) >> 1;
    // this lane's share: band, rows, neighbour pairs
    const int band = (j == 0) ? 0 : (j == 1) ? 1 : (j <= 3) ? 2 : (j == 4) ? 4 : (j == 5) ? 5 : 3;
    const bool split = (j & 2) != 0, first = split && (j & 1) == 0;
    int mband = (band == 0) ? cnt[0] : (band == 1) ? cnt[1] : (band == 2) ? cnt[2] : (band == 3) ? cnt[3] : (band == 4) ? cnt[4] : cnt[5];
    const int h1 = (band == 2) ? h1r : h1i;
    int m = !split ? mband : (first ? h1 : mband - h1);
    if (!fit) { m = 0; mband = 0; }
    const int npairs = m - 1 + ((first && mband > m) ? 1 : 0);     // a first half sees the first row of the second one at slot m
    const int col = lane * STRIDE;
    // destination of every row: low half = slot address, high half = address of its copy at the end of a first half
#pragma unroll
    for (int it = 0; it < CAP; ++it) {
        if ((it & ~(BLK - 1)) < iters) {                       // every trip of a started block of rows
            const int b = code[it] & 0xFF, pos = code[it] >> 8;
            const bool second = (b == 2 && pos >= h1r) || (b == 3 && pos >= h1i);
            const int hb = (b == 2) ? h1r : h1i;
            const int dl = (b == 0) ? 0 : (b == 1) ? 1 : (b == 2) ? 2 : (b == 3) ? 6 : (b == 4) ? 4 : 5;
            const int slot = second ? pos - hb : pos;
            const int dest = (g8 + dl + (second ? 1 : 0)) * STRIDE + slot;
            const int sent = (second && slot == 0) ? dest - STRIDE + hb : 0xFFFF;
            code[it] = (b < 6 && fit) ? (dest | (sent << 16)) : -1;
        }
    }

    // ---- F: fluxes
#pragma unroll
    for (int i0 = 0; i0 < CAP; i0 += BLK) {
        if (i0 < iters) {
            double x[BLK];
#pragma unroll
            for (int q = 0; q < BLK; ++q) {
                const int row = (i0 + q) * 8 + j;
                x[q] = (row < N) ? pf[row] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < BLK; ++q) {
                const int c = code[i0 + q];
                if (c != -1) {
                    buf[c & 0xFFFF] = x[q];
                    if (((c >> 16) & 0xFFFF) != 0xFFFF) buf[(c >> 16) & 0xFFFF] = x[q];
                }
            }
        }
    }
    __syncthreads();
    double v[CAP + 1];
#pragma unroll
    for (int i = 0; i <= CAP; ++i) v[i] = buf[col + i];
    double s = 0.0, mn = __builtin_inf(), mx = -__builtin_inf();
    bool nanf = false;
#pragma unroll
    for (int i = 0; i < CAP; ++i) {
        const bool ok = i < m;
        const double x = v[i];
        s += ok ? x : 0.0;
        nanf = nanf || (ok && is_nan(x));
        mn = dmin(mn, ok ? x : __builtin_inf());
        mx = dmax(mx, ok ? x : -__builtin_inf());
    }
    __syncthreads();

    // ---- T: times; all-rows slope and the order check from the file neighbours
    bool ordered = true, a_snan = false;
    double a_slope = -1.0;
#pragma unroll
    for (int i0 = 0; i0 < CAP; i0 += BLK) {
        if (i0 < iters) {
            double t0[BLK], t1[BLK], f0[BLK], f1[BLK];
#pragma unroll
            for (int q = 0; q < BLK; ++q) {
                const int row = (i0 + q) * 8 + j;
                const bool ok0 = row < N, ok1 = row + 1 < N;
                t0[q] = ok0 ? pt[row] : 0.0;
                t1[q] = ok1 ? pt[row + 1] : 0.0;
                f0[q] = ok0 ? pf[row] : 0.0;
                f1[q] = ok1 ? pf[row + 1] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < BLK; ++q) {
                const int row = (i0 + q) * 8 + j;
                const bool has = row + 1 < N;
                ordered = ordered && !(has && !(t0[q] <= t1[q]));
                const double dt = t1[q] - t0[q];
                const double sl = fabs((f1[q] - f0[q]) / dt);
                const bool valid = has && dt > 0;
                a_snan = a_snan || (valid && is_nan(sl));
                a_slope = (valid && sl > a_slope) ? sl : a_slope;
                const int c = code[i0 + q];
                if (c != -1) {
                    buf[c & 0xFFFF] = t0[q];
                    if (((c >> 16) & 0xFFFF) != 0xFFFF) buf[(c >> 16) & 0xFFFF] = t0[q];
                }
            }
        }
    }
    __syncthreads();
    double slope = -1.0, tmn, tmx;
    bool snan = false;
    {
        double tc = buf[col];
#pragma unroll
        for (int i = 0; i < CAP; ++i) {
            const double tn = buf[col + i + 1];
            const double dt = tn - tc;
            const double sl = fabs((v[i + 1] - v[i]) / dt);
            const bool valid = i < npairs && dt > 0;
            snan = snan || (valid && is_nan(sl));
            slope = (valid && sl > slope) ? sl : slope;
            tc = tn;
        }
        tmn = buf[col];
        tmx = buf[col + ((m > 0) ? m - 1 : 0)];
    }
    __syncthreads();

    // ---- Q: SNR terms (-1 = error bar not positive: the reference leaves the row out)
#pragma unroll
    for (int i0 = 0; i0 < CAP; i0 += BLK) {
        if (i0 < iters) {
            double x[BLK], e[BLK];
#pragma unroll
            for (int q = 0; q < BLK; ++q) {
                const int row = (i0 + q) * 8 + j;
                x[q] = (row < N) ? pf[row] : 0.0;
                e[q] = (row < N) ? pe[row] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < BLK; ++q) {
                const int c = code[i0 + q];
                const double qv = (e[q] > 0) ? fabs(x[q]) / e[q] : -1.0;
                if (c != -1) buf[c & 0xFFFF] = qv;
            }
        }
    }
    __syncthreads();
    double snr = 0.0;
    int nsnr = 0;
#pragma unroll
    for (int i = 0; i < CAP; ++i) {
        const double qv = buf[col + i];
        const bool use = i < m && !(qv < 0);
        snr += use ? qv : 0.0;
        nsnr += use ? 1 : 0;
    }
    __syncthreads();

    // ---- pass-1 totals: per band (the two halves of r and i combined) and over the light curve
    const double sA = G::sum(s), snrA = G::sum(snr);
    const int nsnrA = G::sum(nsnr);
    const bool nanA = G::any(nanf);
    double mnA = G::min(mn), mxA = G::max(mx);
    const double a_slopeA = G::max(a_slope);
    const bool a_snanA = G::any(a_snan);
    const bool orderedA = G::all(ordered);
    double tmnA = G::min((m > 0) ? tmn : __builtin_inf()), tmxA = G::max((m > 0) ? tmx : -__builtin_inf());
    if (split) {
        s += lanes_pair(s);
        snr += lanes_pair(snr);
        nsnr += lanes_pair(nsnr);
        const int p_nan = lanes_pair(nanf ? 1 : 0), p_snan = lanes_pair(snan ? 1 : 0);   // fetched by every lane (no short-circuit)
        nanf = nanf | (p_nan != 0);
        mn = dmin(mn, lanes_pair(mn));
        mx = dmax(mx, lanes_pair(mx));
        const double p_tmn = lanes_pair(tmn), p_tmx = lanes_pair(tmx), p_slope = lanes_pair(slope);
        const int p_m = mband - m;
        tmn = first ? tmn : p_tmn;                     // a band with rows has rows in its first half
        tmx = first ? ((p_m > 0) ? p_tmx : tmx) : ((m > 0) ? tmx : p_tmx);
        slope = (p_slope > slope) ? p_slope : slope;
        snan = snan | (p_snan != 0);
    }
    if (nanf) { mn = qnan(); mx = qnan(); }
    if (nanA) { mnA = qnan(); mxA = qnan(); }
    const double mean = s / mband, meanA = sA / N;

    // ---- pass 2
    double m2 = 0.0, m2A = 0.0;
#pragma unroll
    for (int i = 0; i < CAP; ++i) {
        const bool ok = i < m;
        const double d = v[i] - mean, dA = v[i] - meanA;
        m2 += ok ? d * d : 0.0;
        m2A += ok ? dA * dA : 0.0;
    }
    m2A = G::sum(m2A);
    if (split) m2 += lanes_pair(m2);
    const double sd = (mband > 1) ? sqrt(m2 / mband) : 0.0, sdA = (N > 1) ? sqrt(m2A / N) : 0.0;

    // ---- pass 3 (the lanes of a light curve whose spread is not positive carry garbage that is dropped below)
    double s3 = 0.0, s4 = 0.0, s3A = 0.0, s4A = 0.0;
    int c1 = 0, c2 = 0, c1A = 0, c2A = 0;
    {
        const double inv = 1.0 / sd, sd2 = 2.0 * sd, invA = 1.0 / sdA, sdA2 = 2.0 * sdA;
#pragma unroll
        for (int i = 0; i < CAP; ++i) {
            const bool ok = i < m;
            const double d = v[i] - mean, dA = v[i] - meanA;
            const double zz = d * inv, zA = dA * invA;
            const double z2 = zz * zz, zA2 = zA * zA;
            s3 += ok ? z2 * zz : 0.0;
            s4 += ok ? z2 * z2 : 0.0;
            s3A += ok ? zA2 * zA : 0.0;
            s4A += ok ? zA2 * zA2 : 0.0;
            const double ad = fabs(d), adA = fabs(dA);
            c1 += (ok && ad > sd) ? 1 : 0;
            c2 += (ok && ad > sd2) ? 1 : 0;
            c1A += (ok && adA > sdA) ? 1 : 0;
            c2A += (ok && adA > sdA2) ? 1 : 0;
        }
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

    // ---- order statistics: register sort per lane, then merges across the lanes
    double w[CAP];
#pragma unroll
    for (int i = 0; i < CAP; ++i) w[i] = (i < m) ? v[i] : __builtin_inf();
    reg_sort<CAP>(w);
    double med = qnan(), iqr = qnan(), mad = qnan();
#pragma unroll
    for (int i = 0; i < CAP; ++i) buf[col + i] = w[i];
    __syncthreads();
    if (!split && m > 0 && !nanf) lanes_order_stats<CAP>(buf, col, m, med, iqr, mad);
    __syncthreads();
    lane_merge<CAP, 1>(w, j);
#pragma unroll
    for (int i = 0; i < CAP; ++i) buf[col + i] = w[i];
    __syncthreads();
    if (first && mband > 0 && !nanf) lanes_order_stats<CAP>(buf, col, mband, med, iqr, mad);
    __syncthreads();
    lane_merge<CAP, 2>(w, j);
    lane_merge<CAP, 4>(w, j);
#pragma unroll
    for (int i = 0; i < CAP; ++i) buf[col + i] = w[i];
    __syncthreads();
    double medA = qnan(), iqrA = qnan(), madA = qnan();
    if (j == 3 && N > 0 && !nanA) lanes_order_stats<CAP>(buf, g8 * STRIDE, N, medA, iqrA, madA);
    __syncthreads();
    if (mband <= 1) iqr = 0.0;                               // statistical.py:86: 0 unless the group has two rows
    if (N <= 1) iqrA = 0.0;

    // ---- the 123 columns of every light curve -> LDS rows -> global
    double* o = buf + g * 128;
    if (fit) {
        if (!split || first) {
            double* ob = o + 17 * band;
            if (mband == 0) stat_empty_group(ob, nullptr);
            else stat_write17(ob, mband, mean, sd, mn, mx, med, skew, kurt, mad, iqr, b1, b2, slope, snan, snr, nsnr, tmn, tmx);
        }
        if (j == 3)
            stat_write17(o + 102, N, meanA, sdA, mnA, mxA, medA, skewA, kurtA, madA, iqrA, b1A, b2A, a_slopeA, a_snanA, snrA,
                         nsnrA, tmnA, tmxA);
    }
    __syncthreads();
    if (fit && j == 0) stat_cross_band(o);
    __syncthreads();
    if (fit && orderedA) {
        double* row = out + (int64_t)obj * ld + col0;
#pragma unroll 4
        for (int c = j; c < STAT_NCOL; c += 8) row[c] = o[c];
    } else if (g < nk && j == 0) {
        const int slot = atomicAdd(fallback_count, 1);
        fallback_list[slot] = obj;
    }
    __syncthreads();
}

}  // namespace lcfe
#endif
