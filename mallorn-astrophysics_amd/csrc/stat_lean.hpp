// stat_lean.hpp -- the statistics kernel for the common shape of a light curve: file rows in time
// order, known band codes, at most 512 rows.
//
// The general path (stage.hpp + stat.hpp) keeps a file-order and a band-partitioned copy of the
// samples plus rank-counting scratch in LDS: 19 KiB per 256-row object, i.e. two wavefronts per SIMD,
// and the kernel is latency-bound at that occupancy.  Here the CSR slice goes from global memory
// into registers, is partitioned by band with ballots and lands in LDS once, band-partitioned;
// the all-rows group works on the same copy (its sums and its sort do not care about the order,
// its slopes gather time-neighbours through a 2-byte position map).  10 KiB per 256-row object.
// Objects that do not fit the shape are appended to a list for the general kernel.
#pragma once
#include "stat.hpp"

#if defined(__HIPCC__)
namespace lcfe {

template <int CAP>
struct StatLeanLds {
    double bt[CAP], bf[CAP], be[CAP];   // band-partitioned (u,g,r,i,z,y segments), time-ordered inside a band
    double sorted[CAP];                 // sort scratch: band segments, then all rows
    unsigned short pos_of[CAP];         // file row -> position in bt/bf/be
    double out[STAT_NCOL + 5];
    StatPartial part[6];
    int boff[8];
};

// Registers -> band-partitioned LDS.  Returns false (nothing usable in L) when the object needs the
// general kernel.  Uniform over the wave.
template <int CAP>
__device__ __forceinline__ bool stat_lean_stage(const ObjIn& in, StatLeanLds<CAP>& L) {
    using W = WaveDev;
    constexpr int KPL = CAP / 64;
    const int lane = W::lane(), n = in.n;
    if (n < 1) return false;
    double t[KPL], f[KPL], e[KPL];
    int b[KPL];
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
        const int i = lane + r * 64;
        const bool ok = i < n;
        const int ii = ok ? i : 0;
        t[r] = in.t[ii];
        f[r] = in.f[ii];
        e[r] = in.e[ii];
        b[r] = ok ? (int)in.b[ii] : 256;
    }
    bool ordered = true, known = true;
    int cnt[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
        double nxt = __shfl_down(t[r], 1, 64);
        if (r + 1 < KPL) { const double first = W::rdlane(t[r + 1 < KPL ? r + 1 : r], 0); nxt = (lane == 63) ? first : nxt; }
        const int i = lane + r * 64;
        ordered = ordered && !(i + 1 < n && !(t[r] <= nxt));
        known = known && (b[r] < 6 || b[r] == 256);
#pragma unroll
        for (int k = 0; k < 6; ++k) cnt[k] += popcll(W::ballot(b[r] == k));
    }
    int off[7];
    off[0] = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) off[k + 1] = off[k] + cnt[k];
    if (!W::all(ordered && known)) return false;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) L.boff[k] = off[k];
        L.boff[7] = off[6];
    }
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
        int pos = -1;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const unsigned long long m = W::ballot(b[r] == k);
            if (b[r] == k) pos = off[k] + W::prefix(m);
            off[k] += popcll(m);
        }
        if (pos >= 0) {
            L.bt[pos] = t[r];
            L.bf[pos] = f[r];
            L.be[pos] = e[r];
            L.pos_of[lane + r * 64] = (unsigned short)pos;
        }
    }
    W::sync();
    return true;
}

// All 123 columns of a staged object into L.out.
template <int CAP>
__device__ __forceinline__ void stat_lean_object(int n, StatLeanLds<CAP>& L) {
    using W = WaveDev;
    using WG = GroupDev<8>;
    LCFE_PT0();
    // bands of up to 64 rows side by side in 8-lane groups (4 or 8 values per lane); longer bands one
    // after the other on the full wave
    int mb = 0;
    bool big = false;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int c = L.boff[k + 1] - L.boff[k];
        if (c <= 64) mb = (c > mb) ? c : mb; else big = true;
    }
    const int k = WG::group_id();
    if (k < 6) {
        const int s = L.boff[k], m = L.boff[k + 1] - s;
        double* o = L.out + 17 * k;
        if (m == 0) {
            if (WG::lane() == 0) stat_empty_group(o, &L.part[k]);
        } else if (m > 64) {
        } else if (mb <= 32) {
            group_statistics_fast<WG, 4, false, true>(L.bt + s, L.bf + s, L.be + s, m, L.sorted + s, o, &L.part[k], nullptr);
        } else {
            group_statistics_fast<WG, 8, false, true>(L.bt + s, L.bf + s, L.be + s, m, L.sorted + s, o, &L.part[k], nullptr);
        }
    }
    if (big) {
        W::sync();
        for (int kb = 0; kb < 6; ++kb) {
            const int s = L.boff[kb], m = L.boff[kb + 1] - s;
            double* o = L.out + 17 * kb;
            if (m <= 64) continue;
            if (m <= 128 || CAP <= 128)
                group_statistics_fast<W, 2, false, true>(L.bt + s, L.bf + s, L.be + s, m, L.sorted + s, o, &L.part[kb], nullptr);
            else if (m <= 256 || CAP <= 256)
                group_statistics_fast<W, (CAP > 128) ? 4 : 2, false, true>(L.bt + s, L.bf + s, L.be + s, m, L.sorted + s, o, &L.part[kb], nullptr);
            else
                group_statistics_fast<W, CAP / 64, false, true>(L.bt + s, L.bf + s, L.be + s, m, L.sorted + s, o, &L.part[kb], nullptr);
        }
    }
    W::sync();
    LCFE_PT(1);
    group_statistics_fast<W, CAP / 64, true>(L.bt, L.bf, L.be, n, L.sorted, L.out + 102, nullptr, L.part, L.pos_of);
    LCFE_PT(2);
    W::sync();
    if (W::lane() == 0) stat_cross_band(L.out);
    W::sync();
    LCFE_PT(4);
}

}  // namespace lcfe
#endif
