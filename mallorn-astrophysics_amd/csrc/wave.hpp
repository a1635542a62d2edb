// wave.hpp -- the one-object-per-wavefront execution policy.
//
// Every feature kernel is written as a function template over a policy W that provides the
// wave-level primitives (lane id, reductions, ballot, broadcast, LDS fence).  On the GPU W is
// WaveDev: a 64-lane gfx950 wavefront, one workgroup = one wavefront (so the workgroup barrier
// is the wave-level LDS fence).  tests/hostsim compiles the same templates with WaveHost
// (LANES = 1, reductions are identities) so the arithmetic can be checked on the CPU-only
// build container under sanitizers; that build is test infrastructure and is never loaded by
// the product path.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LCFE_FN __device__ __forceinline__
#define LCFE_FN_NOINLINE __device__ __noinline__
#define LCFE_HD __host__ __device__ inline
#else
#define LCFE_FN inline
#define LCFE_FN_NOINLINE inline
#define LCFE_HD inline
#endif

namespace lcfe {

#if defined(__HIPCC__)
// (DPP moves use __builtin_amdgcn_mov_dpp with bound_ctrl: every pattern used here reads a valid lane, and the
// update_dpp form makes the compiler copy the source into the destination first -- two moves per dword.)
// Value held by lane (l ^ MASK) of the same wavefront, MASK a compile-time constant < 64.  Register
// moves (DPP) where a DPP pattern is an XOR (1, 2, 3 = quad_perm; 7 = row_half_mirror; 8 = row_ror:8;
// 15 = row_mirror), the LDS crossbar without memory traffic otherwise (ds_swizzle inside 32 lanes,
// ds_bpermute across the halves).
template <int MASK>
__device__ __forceinline__ int lane_xor_fetch(int v) {
    static_assert(MASK > 0 && MASK < 64, "lane mask");
    if constexpr (MASK == 1) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);
    else if constexpr (MASK == 2) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);
    else if constexpr (MASK == 3) return __builtin_amdgcn_mov_dpp(v, 0x1B, 0xf, 0xf, true);
    else if constexpr (MASK == 7) return __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true);
    else if constexpr (MASK == 8) return __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, true);
    else if constexpr (MASK == 15) return __builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, true);
    else if constexpr (MASK < 32) return __builtin_amdgcn_ds_swizzle(v, 0x1f | (MASK << 10));
    else return __builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 63) ^ MASK) << 2), v);
}
template <int MASK>
__device__ __forceinline__ double lane_xor_fetch(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = lane_xor_fetch<MASK>((int)b), hi = lane_xor_fetch<MASK>((int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// IEEE minNum / maxNum as single instructions (no NaN reaches the sorting network)
__device__ __forceinline__ double dmin(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double dmax(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#else
inline double dmin(double a, double b) { return (b < a) ? b : a; }
inline double dmax(double a, double b) { return (b > a) ? b : a; }
#endif

#if defined(__HIPCC__)
struct WaveDev {
    static constexpr int LANES = 64;
    static __device__ __forceinline__ int lane() { return threadIdx.x; }
    // LDS fence between cross-lane producer/consumer phases (workgroup == one wave)
    static __device__ __forceinline__ void sync() { __syncthreads(); }
    // sub-group view used by code that lets every WAVEFRONT of a workgroup work redundantly
    static constexpr int WAVE = 64;
    static constexpr int NWAVES = 1;
    static constexpr int NGROUPS = 1;      // independent lane groups (see GroupDev)
    static __device__ __forceinline__ int group_id() { return 0; }
    static __device__ __forceinline__ int wave_id() { return 0; }
    static __device__ __forceinline__ int wlane() { return threadIdx.x; }
    static __device__ __forceinline__ void wave_sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // Reductions: four DPP steps leave every lane with the total of its 16-lane row (quad swaps,
    // row_half_mirror, row_mirror -- register moves, no LDS crossbar), then the four row totals are
    // read with v_readlane and combined; the result is wave-uniform.
    template <int CTRL>
    static __device__ __forceinline__ double dpp(double v) {
        const long long b = __builtin_bit_cast(long long, v);
        int lo = (int)b, hi = (int)(b >> 32);
        lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
        hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
        return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
    }
    template <int CTRL>
    static __device__ __forceinline__ int dpp(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
    static __device__ __forceinline__ double rdlane(double v, int l) {
        const long long b = __builtin_bit_cast(long long, v);
        const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
        return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
    }
    static __device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
    template <class V, class Op>
    static __device__ __forceinline__ V reduce(V v, Op op) {
        v = op(v, dpp<0xB1>(v));      // quad_perm(1,0,3,2)
        v = op(v, dpp<0x4E>(v));      // quad_perm(2,3,0,1)
        v = op(v, dpp<0x141>(v));     // row_half_mirror
        v = op(v, dpp<0x140>(v));     // row_mirror
        return op(op(rdlane(v, 0), rdlane(v, 16)), op(rdlane(v, 32), rdlane(v, 48)));
    }
    static __device__ __forceinline__ double sum(double v) { return reduce(v, [](double a, double b) { return a + b; }); }
    static __device__ __forceinline__ double max(double v) { return reduce(v, [](double a, double b) { return (b > a) ? b : a; }); }
    static __device__ __forceinline__ double min(double v) { return reduce(v, [](double a, double b) { return (b < a) ? b : a; }); }
    static __device__ __forceinline__ int sum(int v) { return reduce(v, [](int a, int b) { return a + b; }); }
    static __device__ __forceinline__ int max(int v) { return reduce(v, [](int a, int b) { return (b > a) ? b : a; }); }
    static __device__ __forceinline__ int min(int v) { return reduce(v, [](int a, int b) { return (b < a) ? b : a; }); }
    static __device__ __forceinline__ unsigned long long ballot(bool p) { return __ballot(p); }
    static __device__ __forceinline__ bool any(bool p) { return __ballot(p) != 0ull; }
    static __device__ __forceinline__ bool all(bool p) { return __ballot(!p) == 0ull; }
    // number of set bits of `mask` strictly below this lane
    static __device__ __forceinline__ int prefix(unsigned long long mask) {
        return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    }
    static __device__ __forceinline__ double bcast(double v, int src) { return __shfl(v, src, 64); }
    static __device__ __forceinline__ int bcast(int v, int src) { return __shfl(v, src, 64); }
    template <int MASK>
    static __device__ __forceinline__ double xfetch(double v) { return lane_xor_fetch<MASK>(v); }
};
#endif

#if defined(__HIPCC__)
// The wavefront of the long-object tier: WaveDev with the object's working set in GLOBAL scratch instead of LDS.  Every
// hand-over between lanes goes through memory the vector caches sit in front of, so every fence -- also the ones that
// are wave-level for LDS -- waits for the stores at workgroup scope; the per-band passes run one after the other on the
// whole wavefront (FitPolicy<LongDev> = LongDev) instead of side by side in lane groups.
struct LongDev : WaveDev {
    static __device__ __forceinline__ void sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    static __device__ __forceinline__ void wave_sync() { sync(); }
};
#endif

#if defined(__HIPCC__)
// GS consecutive lanes of a wavefront working on ONE fit while the other groups of the same wave
// work on other fits (the bands of one light curve).  Groups diverge freely: nothing here uses a
// wave-wide collective or a hardware barrier.  Reductions are DPP moves inside an 8-lane group
// (quad_perm swaps, then row_half_mirror) -- no LDS traffic.
template <int GS>
struct GroupDev {
    static_assert(GS == 8, "DPP pattern below is written for 8-lane groups");
    static constexpr int LANES = GS;
    static constexpr int NGROUPS = 64 / GS;
    static constexpr int WAVE = 64;
    static constexpr int NWAVES = 1;
    static __device__ __forceinline__ int lane() { return threadIdx.x & (GS - 1); }
    static __device__ __forceinline__ int group_id() { return (threadIdx.x & 63) / GS; }
    static __device__ __forceinline__ int wave_id() { return 0; }
    static __device__ __forceinline__ int wlane() { return threadIdx.x & 63; }
    static __device__ __forceinline__ void sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    static __device__ __forceinline__ void wave_sync() { sync(); }
    template <int CTRL>
    static __device__ __forceinline__ double dpp(double v) {
        const long long b = __builtin_bit_cast(long long, v);
        int lo = (int)b, hi = (int)(b >> 32);
        lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
        hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
        return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
    }
    template <int CTRL>
    static __device__ __forceinline__ int dpp(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
    template <int CTRL>
    static __device__ __forceinline__ long long dpp(long long b) {
        const int lo = __builtin_amdgcn_mov_dpp((int)b, CTRL, 0xf, 0xf, true), hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xf, 0xf, true);
        return ((long long)hi << 32) | (unsigned int)lo;
    }
    // quad_perm(1,0,3,2) = 0xB1, quad_perm(2,3,0,1) = 0x4E, row_half_mirror = 0x141
    template <class V, class Op>
    static __device__ __forceinline__ V reduce(V v, Op op) {
        v = op(v, dpp<0xB1>(v));
        v = op(v, dpp<0x4E>(v));
        v = op(v, dpp<0x141>(v));
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
    // ballot over the lanes of this group only (the other groups of the wave may be elsewhere)
    static __device__ __forceinline__ unsigned long long ballot(bool p) {
        return (__ballot(p) >> (group_id() * GS)) & ((1ull << GS) - 1);
    }
    static __device__ __forceinline__ int prefix(unsigned long long mask) {
        return __builtin_popcountll(mask & ((1ull << lane()) - 1));
    }
    template <int MASK>
    static __device__ __forceinline__ double xfetch(double v) {
        static_assert(MASK < GS, "partner outside the lane group");
        return lane_xor_fetch<MASK>(v);
    }
};

// A workgroup of T threads (T/64 wavefronts) working on one object: the GP kernel's policy.
// Reductions go through the wave shuffle network and a small LDS exchange (two barriers each).
template <int T>
struct BlockDev {
    static constexpr int LANES = T;
    static constexpr int NW = T / 64;
    static constexpr int WAVE = 64;
    static constexpr int NWAVES = T / 64;
    static __device__ __forceinline__ int lane() { return threadIdx.x; }
    static __device__ __forceinline__ void sync() { __syncthreads(); }
    static __device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
    static __device__ __forceinline__ int wlane() { return threadIdx.x & 63; }
    // LDS hand-off between lanes of ONE wavefront: LDS operations of a wave execute in order, so
    // only the compiler has to be kept from reordering across this point
    static __device__ __forceinline__ void wave_sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    template <class V, class Op>
    static __device__ __forceinline__ V reduce(V v, Op op) {
        __shared__ V part[NW];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v = op(v, __shfl_xor(v, m, 64));
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
        __syncthreads();
        V r = part[0];
#pragma unroll
        for (int k = 1; k < NW; ++k) r = op(r, part[k]);
        __syncthreads();
        return r;
    }
    static __device__ __forceinline__ double sum(double v) { return reduce(v, [](double a, double b) { return a + b; }); }
    static __device__ __forceinline__ double max(double v) { return reduce(v, [](double a, double b) { return (b > a) ? b : a; }); }
    static __device__ __forceinline__ double min(double v) { return reduce(v, [](double a, double b) { return (b < a) ? b : a; }); }
    static __device__ __forceinline__ int sum(int v) { return reduce(v, [](int a, int b) { return a + b; }); }
    static __device__ __forceinline__ int max(int v) { return reduce(v, [](int a, int b) { return (b > a) ? b : a; }); }
    static __device__ __forceinline__ int min(int v) { return reduce(v, [](int a, int b) { return (b < a) ? b : a; }); }
    static __device__ __forceinline__ bool any(bool p) { return sum(p ? 1 : 0) != 0; }
    static __device__ __forceinline__ bool all(bool p) { return sum(p ? 0 : 1) == 0; }
    // value held (uniformly) by wavefront 0 -> every thread of the workgroup
    static __device__ __forceinline__ double bcast_from_first_wave(double v) {
        __shared__ double slot;
        if (threadIdx.x == 0) slot = v;
        __syncthreads();
        const double r = slot;
        __syncthreads();
        return r;
    }
};
#endif

#if defined(__HIPCC__)
// One wavefront of a larger workgroup working on data of its own while the other wavefronts of the
// workgroup work on theirs (the four bands of the per-band GP): lane ids are wave-relative, every
// "sync" is a wave-level fence, reductions are WaveDev's DPP/readlane ones.
struct WaveOfBlock {
    static constexpr int LANES = 64, WAVE = 64, NWAVES = 1, NGROUPS = 1;
    static __device__ __forceinline__ int lane() { return threadIdx.x & 63; }
    static __device__ __forceinline__ int wlane() { return threadIdx.x & 63; }
    static __device__ __forceinline__ int wave_id() { return 0; }
    static __device__ __forceinline__ int group_id() { return 0; }
    static __device__ __forceinline__ void sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    static __device__ __forceinline__ void wave_sync() { sync(); }
    static __device__ __forceinline__ double sum(double v) { return WaveDev::sum(v); }
    static __device__ __forceinline__ double max(double v) { return WaveDev::max(v); }
    static __device__ __forceinline__ double min(double v) { return WaveDev::min(v); }
    static __device__ __forceinline__ int sum(int v) { return WaveDev::sum(v); }
    static __device__ __forceinline__ int max(int v) { return WaveDev::max(v); }
    static __device__ __forceinline__ int min(int v) { return WaveDev::min(v); }
    static __device__ __forceinline__ unsigned long long ballot(bool p) { return __ballot(p); }
    static __device__ __forceinline__ bool any(bool p) { return __ballot(p) != 0ull; }
    static __device__ __forceinline__ bool all(bool p) { return __ballot(!p) == 0ull; }
    static __device__ __forceinline__ double bcast_from_first_wave(double v) { return v; }
};
#endif

struct WaveHost {
    static constexpr int LANES = 1;
    static constexpr int WAVE = 1;
    static constexpr int NWAVES = 1;
    static constexpr int NGROUPS = 1;
    static int group_id() { return 0; }
    static int lane() { return 0; }
    static void sync() {}
    static int wave_id() { return 0; }
    static int wlane() { return 0; }
    static void wave_sync() {}
    static double sum(double v) { return v; }
    static double max(double v) { return v; }
    static double min(double v) { return v; }
    static int sum(int v) { return v; }
    static int max(int v) { return v; }
    static int min(int v) { return v; }
    static unsigned long long ballot(bool p) { return p ? 1ull : 0ull; }
    static bool any(bool p) { return p; }
    static bool all(bool p) { return p; }
    static int prefix(unsigned long long) { return 0; }
    static double bcast(double v, int) { return v; }
    static int bcast(int v, int) { return v; }
    static double bcast_from_first_wave(double v) { return v; }
    template <int MASK>
    static double xfetch(double v) { return v; }     // never reached: one lane has no partner
};

// Debug builds (-DLCFE_PHASE_PROF): cycle counters per code phase, recorded by thread 0 of a workgroup.
#if defined(LCFE_PHASE_PROF) && defined(__HIPCC__)
__device__ unsigned long long g_phase_prof[32];
// per-workgroup accumulators (same-address global atomics per phase would throttle the kernel)
__device__ __forceinline__ unsigned long long* phase_prof_lds() {
    __shared__ unsigned long long acc[32];
    return acc;
}
#define LCFE_PT0() unsigned long long pt0__ = __builtin_readcyclecounter()
#define LCFE_PT0B() pt0__ = __builtin_readcyclecounter()
#define LCFE_PT(slot_) do { unsigned long long pt1__ = __builtin_readcyclecounter(); if (threadIdx.x == 0) phase_prof_lds()[slot_] += pt1__ - pt0__; pt0__ = __builtin_readcyclecounter(); } while (0)
#define LCFE_PT_INIT() do { if (threadIdx.x < 32) phase_prof_lds()[threadIdx.x] = 0; __syncthreads(); } while (0)
#define LCFE_PT_FLUSH() do { __syncthreads(); if (threadIdx.x < 32) atomicAdd(&g_phase_prof[threadIdx.x], phase_prof_lds()[threadIdx.x]); } while (0)
#else
#define LCFE_PT0() do {} while (0)
#define LCFE_PT0B() do {} while (0)
#define LCFE_PT_INIT() do {} while (0)
#define LCFE_PT_FLUSH() do {} while (0)
#define LCFE_PT(slot_) do {} while (0)
#endif

LCFE_FN int popcll(unsigned long long m) { return __builtin_popcountll(m); }

// quiet NaN without relying on host/device library differences
LCFE_FN double qnan() { return __builtin_nan(""); }
LCFE_FN bool is_nan(double x) { return x != x; }

// Order-preserving map of a double onto uint64 (numpy's sort order: -inf < ... < +inf < NaN).
LCFE_FN uint64_t sort_key(double x) {
    if (x != x) return ~0ull;                       // every NaN sorts last
    uint64_t u = (uint64_t)__builtin_bit_cast(long long, x);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}

// One object's slice of the CSR batch (global memory) + scalars.
struct ObjIn {
    const double* t;
    const double* f;
    const double* e;
    const uint8_t* b;
    int n;
    double z;
};

}  // namespace lcfe
