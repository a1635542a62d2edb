// stat_lanes16.hpp -- the statistics lanes kernel (stat_lanes.hpp) with SIXTEEN lanes per light curve: four light
// curves per wavefront, every band split in time over two lanes (u, g, z, y) or four (r, i):
//
//     lane of the group   0  1   2  3   4  5  6  7   8  9   10 11   12 13 14 15
//     rows                u  u   g  g   r  r  r  r   z  z   y  y    i  i  i  i
//
// A register array of CAP values then holds bands of 2 CAP rows (4 CAP for r and i), and a lane takes 1/16 of the
// rows: light curves of up to 512 rows keep the register and LDS footprint of the 8-lane kernel at up to 256.  Same
// phases and the same arithmetic per element; what differs is the bookkeeping: a band's sums are combined over two
// or four lanes, its order statistics are read after the first (two-lane bands) resp. the second (r, i) merge level,
// the all-rows statistics after the fourth.
#pragma once
#include "stat_lanes.hpp"

#if defined(__HIPCC__)
namespace lcfe {

// reductions over the 16 lanes of a DPP row (= one light curve's group): every lane gets the result
struct Lanes16 {
    template <class V, class Op>
    static __device__ __forceinline__ V reduce(V v, Op op) {
        v = op(v, WaveDev::dpp<0xB1>(v));      // quad_perm(1,0,3,2)
        v = op(v, WaveDev::dpp<0x4E>(v));      // quad_perm(2,3,0,1)
        v = op(v, WaveDev::dpp<0x141>(v));     // row_half_mirror
        v = op(v, WaveDev::dpp<0x140>(v));     // row_mirror
        return v;
    }
    static __device__ __forceinline__ double sum(double v) { return reduce(v, [](double a, double b) { return a + b; }); }
    static __device__ __forceinline__ double max(double v) { return reduce(v, [](double a, double b) { return (b > a) ? b : a; }); }
    static __device__ __forceinline__ double min(double v) { return reduce(v, [](double a, double b) { return (b < a) ? b : a; }); }
    static __device__ __forceinline__ int sum(int v) { return reduce(v, [](int a, int b) { return a + b; }); }
    static __device__ __forceinline__ int max(int v) { return reduce(v, [](int a, int b) { return (b > a) ? b : a; }); }
    static __device__ __forceinline__ int min(int v) { return reduce(v, [](int a, int b) { return (b < a) ? b : a; }); }
    static __device__ __forceinline__ bool any(bool p) { return max(p ? 1 : 0) != 0; }
    static __device__ __forceinline__ bool all(bool p) { return min(p ? 1 : 0) != 0; }
    static __device__ __forceinline__ void sync() { GroupDev<8>::sync(); }
};

// op over the lanes of one band: the lane pair, and for r and i (`four`) also the other pair of the quad
template <class V, class Op>
__device__ __forceinline__ V lanes16_band(V x, bool four, Op op) {
    const V x1 = op(x, lane_xor_fetch<1>(x));
    const V x2 = op(x1, lane_xor_fetch<2>(x1));
    return four ? x2 : x1;
}

// Four light curves (one per 16-lane group: `obj` < 0 = none; CSR rows [s1, e1)) -> their 123 columns, or list
// `fallback_list`.  ITERS = rows / 16 a light curve of the list may have.
template <int CAP, int ITERS>
__device__ __forceinline__ void stat_lanes16_batch(const double* gt, const double* gf, const double* ge, const uint8_t* gb, int obj,
                                                   int64_t s1, int64_t e1, LanesBuf buf, LanesBuf all_rows, double* out, int ld,
                                                   int col0, int* fallback_list, int* fallback_count) {
    using G = Lanes16;
    constexpr int STRIDE = StatLanesLds<CAP>::STRIDE;
    constexpr int BLK = 4;
    static_assert(ITERS % BLK == 0 && ITERS <= CAP, "rows per lane");
    const int lane = threadIdx.x & 63, j = lane & 15, g16 = lane & 48;
    const int n = (int)(e1 - s1);
    const bool has_obj = obj >= 0;
    bool fit = has_obj && n >= 1 && n <= 16 * ITERS;
    const int nr = fit ? n : 0;
    int nmax = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int v_ = __builtin_amdgcn_readlane(nr, 16 * k); nmax = (v_ > nmax) ? v_ : nmax; }
    const int iters = (nmax + 15) >> 4;
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
                    const int row = (i0 + q) * 16 + j;
                    const bool ok = row < nr;
                    code[i0 + q] = ok ? (int)pb[row] : 256;
                    rt[i0 + q] = ok ? pt[row] : 0.0;
                    rf[i0 + q] = ok ? pf[row] : 0.0;
                    rq[i0 + q] = ok ? pe[row] : 0.0;
                }
            }
        }
    }

    // ---- A: slot of every row inside its band = rows of that band before it (ballots over the 16 rows of a trip)
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
                const unsigned int m16 = (unsigned int)(__ballot(b == k) >> g16) & 0xFFFFu;
                if (b == k) { mine = m16; c0 = cnt[k]; }
                cnt[k] += __builtin_popcount(m16);
            }
            const int pos = c0 + __builtin_popcount(mine & ((1u << j) - 1u));
            code[it] = ((b < 6) ? b : 7) | (pos << 8);
        }
    }
    fit = fit && G::all(known) && cnt[0] <= 2 * CAP && cnt[1] <= 2 * CAP && cnt[4] <= 2 * CAP && cnt[5] <= 2 * CAP &&
          cnt[2] <= 4 * CAP && cnt[3] <= 4 * CAP;
    const int N = fit ? n : 0;
    // rows per part of every band (parts of a band: 2, r and i: 4)
    int hb[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) hb[k] = (k == 2 || k == 3) ? (cnt[k] + 3) >> 2 : (cnt[k] + 1) >> 1;
    // this lane's share: band, part, rows
    const int band = (j < 2) ? 0 : (j < 4) ? 1 : (j < 8) ? 2 : (j < 10) ? 4 : (j < 12) ? 5 : 3;
    const bool four = (band == 2 || band == 3);
    const int part = four ? (j & 3) : (j & 1), nparts = four ? 4 : 2;
    int mband = (band == 0) ? cnt[0] : (band == 1) ? cnt[1] : (band == 2) ? cnt[2] : (band == 3) ? cnt[3] : (band == 4) ? cnt[4] : cnt[5];
    const int h = (band == 0) ? hb[0] : (band == 1) ? hb[1] : (band == 2) ? hb[2] : (band == 3) ? hb[3] : (band == 4) ? hb[4] : hb[5];
    int m = mband - part * h;
    m = (m < 0) ? 0 : ((m > h) ? h : m);
    int m_next = mband - (part + 1) * h;
    m_next = (part + 1 < nparts && m_next > 0) ? 1 : 0;
    if (!fit) { m = 0; mband = 0; m_next = 0; }
    const bool bridge = m_next != 0;                        // the pair (last row of this part, first row of the next)
    const bool first = part == 0;
    const int col = lane * STRIDE;

    // ---- per row: LDS slot (-1: none); the all-rows slope and the order check from the file neighbours; the SNR term
    bool ordered = true, a_snan = false;
    double a_slope = -1.0;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        if ((it & ~(BLK - 1)) < iters) {
            const int b = code[it] & 0xFF, pos = code[it] >> 8;
            const int hh = (b == 0) ? hb[0] : (b == 1) ? hb[1] : (b == 2) ? hb[2] : (b == 3) ? hb[3] : (b == 4) ? hb[4] : hb[5];
            const int jb = (b == 0) ? 0 : (b == 1) ? 2 : (b == 2) ? 4 : (b == 3) ? 12 : (b == 4) ? 8 : 10;
            const int p = ((pos >= hh) ? 1 : 0) + ((pos >= 2 * hh) ? 1 : 0) + ((pos >= 3 * hh) ? 1 : 0);   // (two parts: pos < 2 hh)
            code[it] = (b < 6 && fit) ? (g16 + jb + p) * STRIDE + (pos - p * hh) : -1;
            const int row = it * 16 + j;
            const bool has = row + 1 < N;
            const double tn_l = WaveDev::dpp<0x101>(rt[it]), fn_l = WaveDev::dpp<0x101>(rf[it]);       // row_shl:1
            const double tn_w = WaveDev::dpp<0x11F>(rt[(it + 1 < ITERS) ? it + 1 : it]),                // row_shr:15
                         fn_w = WaveDev::dpp<0x11F>(rf[(it + 1 < ITERS) ? it + 1 : it]);
            const double t1 = (j == 15) ? tn_w : tn_l, f1 = (j == 15) ? fn_w : fn_l;
            ordered = ordered && !(has && !(rt[it] <= t1));
            const double dt = t1 - rt[it];
            const double sl = stat_slope(f1 - rf[it], dt);
            const bool valid = has && dt > 0;
            a_snan = a_snan || (valid && is_nan(sl));
            a_slope = (valid && sl > a_slope) ? sl : a_slope;
            rq[it] = (rq[it] > 0) ? fabs(rf[it]) / rq[it] : -0.0;
        }
    }

    // ---- F: fluxes -> lanes (cleared columns: the slots behind a lane's rows read as 0.0 in every phase)
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
    const double f_last = buf[col + ((m > 0) ? m - 1 : 0)], f_next = buf[col + ((lane < 63) ? STRIDE : 0)];
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

    // ---- pass-1 totals: per band (its two or four lanes combined) and over the light curve
    const double sA = G::sum(s), snrA = G::sum(snr);
    const int nsnrA = G::sum(nsnr);
    const bool nanA = G::any(nanf);
    const double a_slopeA = G::max(a_slope);
    const bool a_snanA = G::any(a_snan);
    const bool orderedA = G::all(ordered);
    const double tmnA = G::min((m > 0) ? tmn : __builtin_inf()), tmxA = G::max((m > 0) ? tmx : -__builtin_inf());
    auto add = [](auto a, auto b) { return a + b; };
    auto mxo = [](auto a, auto b) { return (b > a) ? b : a; };
    auto mno = [](auto a, auto b) { return (b < a) ? b : a; };
    s = lanes16_band(s, four, add);
    snr = lanes16_band(snr, four, add);
    nsnr = lanes16_band(nsnr, four, add);
    nanf = lanes16_band(nanf ? 1 : 0, four, mxo) != 0;
    snan = lanes16_band(snan ? 1 : 0, four, mxo) != 0;
    slope = lanes16_band(slope, four, mxo);
    tmn = lanes16_band((m > 0) ? tmn : __builtin_inf(), four, mno);
    tmx = lanes16_band((m > 0) ? tmx : -__builtin_inf(), four, mxo);
    const double mean = s / mband, meanA = sA / N;
    LanesBuf oa = all_rows + (g16 >> 4) * 17;              // the all-rows columns leave the registers as soon as they are known
    if (j == 1) {
        oa[0] = (double)N;
        oa[1] = meanA;
        oa[13] = (N > 1) ? (a_snanA ? qnan() : (a_slopeA < 0 ? 0.0 : a_slopeA)) : 0.0;
        oa[14] = (nsnrA > 0) ? snrA / nsnrA : qnan();
        oa[15] = (N > 1) ? (tmxA - tmnA) : 0.0;
        oa[16] = (N > 1) ? (tmxA - tmnA) / (double)(N - 1) : 0.0;
    }

    // ---- behind the lane's rows the band mean: the centred band passes need no per-element mask
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
    m2 = lanes16_band(m2, four, add);
    const double sd = (mband > 1) ? sqrt(m2 / mband) : 0.0, sdA = (N > 1) ? sqrt(m2A / N) : 0.0;

    // ---- pass 3
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
    s3 = lanes16_band(s3, four, add);
    s4 = lanes16_band(s4, four, add);
    c1 = lanes16_band(c1, four, add);
    c2 = lanes16_band(c2, four, add);
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
    if (j == 1) {
        oa[2] = sdA;
        oa[6] = skewA;
        oa[7] = kurtA;
        oa[11] = b1A;
        oa[12] = b2A;
    }

    // ---- order statistics and extrema: register sort per lane, then merges across the lanes; a sequence's extrema
    //      and order statistics are read off the LDS dump of the level that completes it
    double w[CAP];
#pragma unroll
    for (int i = 0; i < CAP; ++i) w[i] = (i < m) ? v[i] : __builtin_inf();
    reg_sort<CAP>(w);
    double mn = qnan(), mx = qnan(), med = qnan(), iqr = qnan(), mad = qnan();
    lane_merge<CAP, 1>(w, j);
#pragma unroll
    for (int i = 0; i < CAP; ++i) buf[col + i] = w[i];
    G::sync();
    if (!four && first && mband > 0 && !nanf) {
        mn = buf[col];
        mx = lanes_seq<CAP>(buf, col, mband - 1);
        lanes_order_stats<CAP>(buf, col, mband, med, iqr, mad);
    }
    G::sync();
    lane_merge<CAP, 2>(w, j);
#pragma unroll
    for (int i = 0; i < CAP; ++i) buf[col + i] = w[i];
    G::sync();
    if (four && first && mband > 0 && !nanf) {
        mn = buf[col];
        mx = lanes_seq<CAP>(buf, col, mband - 1);
        lanes_order_stats<CAP>(buf, col, mband, med, iqr, mad);
    }
    G::sync();
    lane_merge<CAP, 4>(w, j);
    lane_merge<CAP, 8>(w, j);
#pragma unroll
    for (int i = 0; i < CAP; ++i) buf[col + i] = w[i];
    G::sync();
    if (j == 1) {
        double mnA = qnan(), mxA = qnan(), medA = qnan(), iqrA = qnan(), madA = qnan();
        if (N > 0 && !nanA) {
            mnA = buf[g16 * STRIDE];
            mxA = lanes_seq<CAP>(buf, g16 * STRIDE, N - 1);
            lanes_order_stats<CAP>(buf, g16 * STRIDE, N, medA, iqrA, madA);
        }
        if (N <= 1) iqrA = 0.0;
        oa[3] = mnA;
        oa[4] = mxA;
        oa[8] = mxA - mnA;
        oa[5] = medA;
        oa[9] = madA;
        oa[10] = iqrA;
    }
    G::sync();
    if (mband <= 1) iqr = 0.0;                               // statistical.py:86: 0 unless the group has two rows

    // ---- the 123 columns of every light curve -> LDS rows -> global
    LanesBuf o = buf + (g16 >> 4) * 128;
    if (fit && first) {
        double* ob = lanes_raw(o + 17 * band, 17);
        if (mband == 0) stat_empty_group(ob, nullptr);
        else stat_write17(ob, mband, mean, sd, mn, mx, med, skew, kurt, mad, iqr, b1, b2, slope, snan, snr, nsnr, tmn, tmx);
    }
    G::sync();
    if (fit) {
#pragma unroll
        for (int c = j; c < 17; c += 16) o[102 + c] = oa[c];
    }
    G::sync();
    if (fit && j == 0) stat_cross_band(lanes_raw(o, STAT_NCOL));
    G::sync();
    const bool done = fit && orderedA;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int obj_r = __builtin_amdgcn_readlane(done ? obj : -1, 16 * r);
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

// One workgroup = one batch (`batch`) of four consecutive list entries; `buf`: 64 x (CAP + 1) doubles of LDS.
template <int CAP, int ITERS>
__device__ __forceinline__ void stat_lanes16_run(const int64_t* offsets, const double* gt, const double* gf, const double* ge,
                                                 const uint8_t* gb, const int* list, int count, int batch, LanesBuf buf, LanesBuf all_rows,
                                                 double* out, int ld, int col0, int* fallback_list, int* fallback_count) {
    const int g = (threadIdx.x & 63) >> 4;
    const int64_t base = (int64_t)batch * 4;
    int obj = -1;
    int64_t s1 = 0, e1 = 0;
    if (base + g < count) {
        obj = list[base + g];
        s1 = offsets[obj];
        e1 = offsets[obj + 1];
    }
    stat_lanes16_batch<CAP, ITERS>(gt, gf, ge, gb, obj, s1, e1, buf, all_rows, out, ld, col0, fallback_list, fallback_count);
}

}  // namespace lcfe
#endif
