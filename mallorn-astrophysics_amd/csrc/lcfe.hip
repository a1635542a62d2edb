// lcfe.hip -- gfx950 kernels + C-ABI of liblcfe.so (see include/lcfe.h).
//
// Execution model: one light curve per 64-lane wavefront, one wavefront per workgroup, a
// persistent grid that strides over the objects.  Per feature set the object's working set
// (staged samples, sorted views, Jacobians, Gram matrices) lives in LDS; HBM is touched once for
// the CSR slice and once for the output row.  Objects are binned into LDS tiers by point count
// (CAP = 128 .. 2048) by one small kernel per call that writes an index list per tier; each tier is
// one launch over its list, so short light curves run at high occupancy and long ones still fit.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lcfe.h"
#include "feature_sets.hpp"
#include "stat_lean.hpp"
#include "stat_lanes.hpp"
#include "stat_lanes16.hpp"
#include "gp1d.hpp"

using namespace lcfe;

namespace {

thread_local std::string g_err;

// Restores the caller's current device on every exit path of an entry point that had to switch devices.
struct DeviceGuard {
    int prev = -1;
    bool armed = false;
    int enter(int device) {
        if (device < 0) return 0;
        if (hipGetDevice(&prev) != hipSuccess) return 1;
        if (prev == device) return 0;
        if (hipSetDevice(device) != hipSuccess) return 1;
        armed = true;
        return 0;
    }
    ~DeviceGuard() { if (armed) (void)hipSetDevice(prev); }
};

int fail(const char* what, hipError_t e, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_err = buf;
    return 1;
}
int fail_msg(const std::string& m) {
    g_err = m;
    return 2;
}
#define HIP_TRY(expr)                                                  \
    do {                                                               \
        hipError_t e_ = (expr);                                        \
        if (e_ != hipSuccess) return fail(#expr, e_, __FILE__, __LINE__); \
    } while (0)

struct BatchView {
    const int64_t* offsets;
    const double* t;
    const double* f;
    const double* e;
    const uint8_t* b;
    const double* z;
    int64_t n_obj;
};

// ---- binning: index lists per LDS tier (feature sets) and per Gram-matrix tier (GP)
constexpr int kNumBins = 7;            // up to six tiers + "longer than the largest tier" (bin 6; the sets use bins 0..4 + 6)
constexpr int kBinThreads = 1024;
constexpr int kNumLists = 2 * kNumBins + 12 + 6;
constexpr int kStatFallbackList = 2 * kNumBins;   // objects the lean statistics kernel hands to the general one
constexpr int kBazinFallbackList = 2 * kNumBins + 1;   // objects with a band longer than the largest fit tier
constexpr int kPowerlawFallbackList = 2 * kNumBins + 2;
constexpr int kStatRetryList = 2 * kNumBins + 3;      // light curves of up to 512 rows the lanes kernels do not take (stat_plan_kernel)
constexpr int kStatL16List = 2 * kNumBins + 4;        // light curves of up to 128 rows whose bands fit 16-row lanes (r, i: 32 rows)
constexpr int kStatL32List = 2 * kNumBins + 5;        // ... 32-row lanes (r, i: 64 rows)
constexpr int kStatL32xList = 2 * kNumBins + 6;       // light curves of up to 256 rows whose bands fit 32-row lanes
constexpr int kStatW16List = 2 * kNumBins + 7;        // ... up to 256 rows, 32-row lanes with 16 lanes per light curve (bands of up to 64 rows, r, i: 128)
constexpr int kStatW32List = 2 * kNumBins + 8;        // ... up to 512 rows, the same
constexpr int kBazinLongList = 2 * kNumBins + 9;      // light curves of more than 1024 rows with a band beyond the largest fit tier
constexpr int kPowerlawLongList = 2 * kNumBins + 10;  // ... with more post-peak rows in a band than the largest fit tier
constexpr int kResearchLongList = 2 * kNumBins + 11;  // light curves whose r band spans more days than the Mexican-hat grid in LDS
constexpr int kGpSortedList = 2 * kNumBins + 12;      // + tier (0..5): the 2-D GP tier's light curves, longest first (gp_sort_kernel)
struct Bins {
    int* lists;                        // [kNumLists][n_obj]: set tiers, GP tiers, statistics fallback
    int* counts;                       // [kNumLists]
    int64_t stride;                    // n_obj
};

__device__ __forceinline__ int set_bin_of(int64_t n) {
    return (n <= 128) ? 0 : (n <= 256) ? 1 : (n <= 512) ? 2 : (n <= 1024) ? 3 : (n <= 2048) ? 4 : 6;
}
constexpr int kGpSmallNP = 240;     // matrix in global scratch, two pivot tiles per pass, two 512-thread workgroups per CU
constexpr int kGpMidNP = 512;       // matrix in global scratch, two pivot tiles per pass, one 512-thread workgroup per CU
constexpr int kGpGlobalNP = 768;    // matrix in global scratch, one pivot tile per pass (512..767 rows)
// GP window on the ROW count of the object (>= its valid points); one row of the tile storage is the
// augmented residual row, so the caps are NP - 1.
__device__ __forceinline__ int gp_bin_of(int64_t n) {
    return (n <= 63) ? 0 : (n <= 111) ? 1 : (n <= 159) ? 2 : (n <= kGpSmallNP - 1) ? 3 : (n <= kGpMidNP - 1) ? 4 : (n <= kGpGlobalNP - 1) ? 5 : 6;
}

// One block bins 1024 consecutive objects: ballots give the rank inside a wave, an LDS scan the
// wave's offset inside the block, one atomicAdd per (block, bin) the block's slice of the list --
// so a list keeps file order inside every 1024-object block (CSR locality for the tier kernels).
__global__ __launch_bounds__(kBinThreads) void bin_kernel(const int64_t* offsets, int64_t n_obj, int* lists, int* counts) {
    __shared__ int wcount[kBinThreads / 64][2 * kNumBins];
    __shared__ int base[2 * kNumBins];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * kBinThreads + threadIdx.x;
    int bin[2] = {-1, -1}, rank[2] = {0, 0};
    if (i < n_obj) {
        const int64_t n = offsets[i + 1] - offsets[i];
        bin[0] = set_bin_of(n);
        bin[1] = gp_bin_of(n);
    }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int b = 0; b < kNumBins; ++b) {
            const unsigned long long m = __ballot(bin[g] == b);
            if (bin[g] == b) rank[g] = WaveDev::prefix(m);
            if (lane == 0) wcount[wave][g * kNumBins + b] = popcll(m);
        }
    __syncthreads();
    if (threadIdx.x < 2 * kNumBins) {
        int total = 0;
        for (int w = 0; w < kBinThreads / 64; ++w) {
            const int c = wcount[w][threadIdx.x];
            wcount[w][threadIdx.x] = total;
            total += c;
        }
        base[threadIdx.x] = total ? atomicAdd(&counts[threadIdx.x], total) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 2; ++g)
        if (bin[g] >= 0) {
            const int k = g * kNumBins + bin[g];
            lists[(int64_t)k * n_obj + base[k] + wcount[wave][k] + rank[g]] = (int)i;
        }
}

// NaN rows + status -100 for the objects of bins [from, kNumBins) of group `g` (too long for the
// tiers this launch sequence covers); every block of the last tier's launch takes a share.
template <class W>
__device__ void nan_fill_bins(const Bins& bins, int g, int from, double* out, int ld, int col0, int ncol,
                              int32_t* status, int st_ld, int st0, int nst) {
    for (int b = from; b < kNumBins; ++b) {
        const int c = bins.counts[g * kNumBins + b];
        const int* list = bins.lists + (int64_t)(g * kNumBins + b) * bins.stride;
        for (int pos = blockIdx.x; pos < c; pos += gridDim.x) {
            const int64_t i = list[pos];
            fill_row_nan<W>(out + i * (int64_t)ld + col0, ncol);
            if (status && nst)
                for (int k = threadIdx.x; k < nst; k += blockDim.x) status[i * (int64_t)st_ld + st0 + k] = -100;
        }
    }
}

// One launch = one (feature set, LDS tier): the objects of the tier's index list are handed out
// through a device-side ticket counter (zeroed once per call) -- fit costs are heavy-tailed (nfev
// 5..2000), so a static round-robin would leave most waves idle behind the unluckiest one.  One
// ticket = `chunk` consecutive list entries (a single counter saturates near 90 tickets/us, so the
// cheap streaming sets take 8 objects per ticket; the fits take one); the chunk's list entries and
// CSR offsets are fetched by its first lanes in one go.  The launch of the last tier also writes the
// NaN rows of the objects that are too long for it (bins >= nan_from).
// minimum waves per SIMD the register allocation must leave room for: the bounded fits are long
// dependent fp64 chains, so a second wave per SIMD matters more than keeping every value in a register
// (only the 128-row tier: the larger tiers are limited to one wave per SIMD by their LDS footprint anyway)
template <int SET, int CAP> struct set_waves { static constexpr int N = ((SET == SET_BAZIN || SET == SET_POWERLAW) && CAP <= 128) ? 2 : 1; };

template <int SET, int CAP>
__global__ __launch_bounds__(64, (set_waves<SET, CAP>::N)) void set_kernel(BatchView B, Bins bins, int bin, int nan_from, double* out,
                                                 int ld, int col0, int32_t* status, int st_ld,
                                                 int st0, unsigned long long* ticket, int chunk) {
    __shared__ SetLds<SET, CAP> ws;
    __shared__ long long next_ticket;
    using W = WaveDev;
    const int ncol = set_ncols(SET);
    const int nst = set_nstatus(SET);
    const int count = bins.counts[bin];
    const int* list = bins.lists + (int64_t)bin * bins.stride;
    // a sparsely filled tier hands out smaller tickets, so that every wave gets several of them
    const int per_wave = count / (4 * (int)gridDim.x);
    chunk = (per_wave < 1) ? 1 : ((per_wave < chunk) ? per_wave : chunk);
    LCFE_PT_INIT();
    LCFE_PT0();
    for (;;) {
        if (threadIdx.x == 0) next_ticket = (long long)atomicAdd(ticket, 1ull);
        __syncthreads();
        const int64_t base = next_ticket * chunk;
        __syncthreads();
        if (base >= count) break;
        const int nk = (count - base < chunk) ? (int)(count - base) : chunk;
        int obj_l = 0, n_l = 0;
        long long s_l = 0;
        if ((int)threadIdx.x < nk) {
            obj_l = list[base + threadIdx.x];
            s_l = B.offsets[obj_l];
            n_l = (int)(B.offsets[obj_l + 1] - s_l);
        }
        for (int k = 0; k < nk; ++k) {
            const int64_t i = W::rdlane(obj_l, k);
            const int64_t s = ((int64_t)W::rdlane((int)(s_l >> 32), k) << 32) | (unsigned int)W::rdlane((int)s_l, k);
            const int n = W::rdlane(n_l, k);
            double* row = out + i * (int64_t)ld + col0;
            int32_t* st = (status && nst) ? status + i * (int64_t)st_ld + st0 : nullptr;
            ObjIn in{B.t + s, B.f + s, B.e + s, B.b + s, n, B.z ? B.z[i] : qnan()};
            LCFE_PT(5);
            const int rc = RunSet<W, SET, CAP>::run(in, ws, row, st);
            if constexpr (SET == SET_RESEARCH) {
                // r band longer than the Mexican-hat grid of this tier: the long-object tier takes the light curve
                if (rc == -100 && threadIdx.x == 0) {
                    const int slot = atomicAdd(&bins.counts[kResearchLongList], 1);
                    bins.lists[(int64_t)kResearchLongList * bins.stride + slot] = (int)i;
                }
            }
            (void)rc;
            LCFE_PT0B();
        }
    }
    nan_fill_bins<W>(bins, 0, nan_from, out, ld, col0, ncol, status, st_ld, st0, nst);
    LCFE_PT(5);
    LCFE_PT_FLUSH();
}

// Statistics, lean layout (stat_lean.hpp) for the tiers up to 512 rows.  Same ticket scheme as
// set_kernel; an object that does not fit the lean shape is appended to the fallback list, which a
// set_kernel<SET_STAT, CAP> launch processes afterwards.
// waves per SIMD the LDS footprint allows (5.6 / 9.9 / 18.4 KiB per object): the register budget follows
template <int CAP> struct stat_lean_waves { static constexpr int N = (CAP <= 128) ? 4 : ((CAP <= 256) ? 3 : 2); };

template <int CAP>
__global__ __launch_bounds__(64, stat_lean_waves<CAP>::N) void stat_lean_kernel(BatchView B, Bins bins, int bin, double* out, int ld, int col0,
                                                       unsigned long long* ticket, int chunk) {
    __shared__ StatLeanLds<CAP> L;
    __shared__ long long next_ticket;
    using W = WaveDev;
    const int count = bins.counts[bin];
    const int* list = bins.lists + (int64_t)bin * bins.stride;
    // tickets of at least two light curves: one counter serves about 90 tickets per microsecond, which a statistics
    // tier of 20 000 light curves would otherwise spend more time on than on the light curves
    const int per_wave = count / (2 * (int)gridDim.x);
    chunk = (per_wave < 2) ? 2 : ((per_wave < chunk) ? per_wave : chunk);
    LCFE_PT_INIT();
    LCFE_PT0();
    for (;;) {
        if (threadIdx.x == 0) next_ticket = (long long)atomicAdd(ticket, 1ull);
        __syncthreads();
        const int64_t base = next_ticket * chunk;
        __syncthreads();
        if (base >= count) break;
        const int nk = (count - base < chunk) ? (int)(count - base) : chunk;
        int obj_l = 0, n_l = 0;
        long long s_l = 0;
        if ((int)threadIdx.x < nk) {
            obj_l = list[base + threadIdx.x];
            s_l = B.offsets[obj_l];
            n_l = (int)(B.offsets[obj_l + 1] - s_l);
        }
        for (int k = 0; k < nk; ++k) {
            const int obj = W::rdlane(obj_l, k);
            const int64_t s = ((int64_t)W::rdlane((int)(s_l >> 32), k) << 32) | (unsigned int)W::rdlane((int)s_l, k);
            const int n = W::rdlane(n_l, k);
            ObjIn in{B.t + s, B.f + s, B.e + s, B.b + s, n, qnan()};
            LCFE_PT(5);
            const bool lean = stat_lean_stage<CAP>(in, L);
            LCFE_PT(0);
            if (lean) {
                stat_lean_object<CAP>(n, L);
                LCFE_PT0B();
                store_row<W>(L.out, out + (int64_t)obj * ld + col0, STAT_NCOL);
                W::sync();
                LCFE_PT(3);
            } else if (threadIdx.x == 0) {
                const int slot = atomicAdd(&bins.counts[kStatFallbackList], 1);
                bins.lists[(int64_t)kStatFallbackList * bins.stride + slot] = obj;
            }
        }
    }
    LCFE_PT(5);
    LCFE_PT_FLUSH();
}

// ---------------------------------------------------------------------------------------------------------------
// The long-object tier.  The reference has no limit on the rows of a light curve; the LDS tiers end at 2048 rows
// (1024 for the object-level fits and the research set).  Longer light curves -- and the ones that overflow a per-band
// limit of the LDS tiers -- run through the SAME templates with CAP = kLongCap, their working set (the structures that
// are LDS in the tiers) in a per-workgroup slab of global scratch: one wavefront per workgroup, a handful of workgroups
// (such objects are rare; the reference's own loaders truncate light curves at 300-500 points).  Up to three index
// lists are drained through one ticket counter.  Light curves beyond kLongCap rows keep NaN and status -100.
constexpr int kLongCap = 16384;
constexpr int kLongGrid = 16;
template <int SET> constexpr size_t long_slab_bytes() { return (sizeof(SetLds<SET, kLongCap>) + 255) & ~(size_t)255; }

template <int SET>
__global__ __launch_bounds__(64) void set_long_kernel(BatchView B, Bins bins, int l0, int l1, int l2, double* out, int ld, int col0,
                                                      int32_t* status, int st_ld, int st0, unsigned long long* ticket, char* slabs) {
    using W = LongDev;
    __shared__ long long next_ticket;
    SetLds<SET, kLongCap>& ws = *reinterpret_cast<SetLds<SET, kLongCap>*>(slabs + (size_t)blockIdx.x * long_slab_bytes<SET>());
    const int ncol = set_ncols(SET), nst = set_nstatus(SET);
    const int c0 = (l0 >= 0) ? bins.counts[l0] : 0, c1 = (l1 >= 0) ? bins.counts[l1] : 0, c2 = (l2 >= 0) ? bins.counts[l2] : 0;
    for (;;) {
        if (threadIdx.x == 0) next_ticket = (long long)atomicAdd(ticket, 1ull);
        __syncthreads();
        const int64_t pos = next_ticket;
        __syncthreads();
        if (pos >= (int64_t)c0 + c1 + c2) break;
        const int li = (pos < c0) ? l0 : ((pos < (int64_t)c0 + c1) ? l1 : l2);
        const int64_t at = (pos < c0) ? pos : ((pos < (int64_t)c0 + c1) ? pos - c0 : pos - c0 - c1);
        // (wave-uniform by construction; said explicitly, so that the branch on the length below is a scalar branch --
        //  as a divergent `continue` the compiler turned it into an inner loop that re-read the same ticket for ever)
        const int64_t i = __builtin_amdgcn_readfirstlane(bins.lists[(int64_t)li * bins.stride + at]);
        const int64_t s = B.offsets[i];
        const int64_t n = B.offsets[i + 1] - s;
        double* row = out + i * (int64_t)ld + col0;
        int32_t* st = (status && nst) ? status + i * (int64_t)st_ld + st0 : nullptr;
        const bool fits = __builtin_amdgcn_readfirstlane((int)(n <= kLongCap)) != 0;
        if (fits) {
            ObjIn in{B.t + s, B.f + s, B.e + s, B.b + s, (int)n, B.z ? B.z[i] : qnan()};
            RunSet<W, SET, kLongCap>::run(in, ws, row, st);
        } else {
            fill_row_nan<W>(row, ncol);
            if (st) for (int k = threadIdx.x; k < nst; k += 64) st[k] = -100;
        }
        __syncthreads();
    }
}

template <int SET>
int launch_long(const BatchView& B, const Bins& bins, int l0, int l1, int l2, double* out, int ld, int col0, int32_t* status, int st_ld,
                int st0, hipStream_t stream, unsigned long long* ticket, char* slabs) {
    if (!slabs) return 0;
    hipLaunchKernelGGL(set_long_kernel<SET>, dim3(kLongGrid), dim3(64), 0, stream, B, bins, l0, l1, l2, out, ld, col0, status, st_ld, st0,
                       ticket, slabs);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Statistics, eight light curves per wavefront and one lane per band (stat_lanes.hpp: CAP = rows of u, g, z, y a lane
// holds -- r and i: 2 CAP over two lanes; ITERS = rows / 8 of the light curve).  One workgroup = one batch of eight
// consecutive entries of one of the lists the plan kernels fill: 32-row lanes for light curves of up to 256 rows, 32-
// and 16-row lanes for those of up to 128 rows -- or one batch of FOUR entries of the two lists of light curves with
// longer bands (stat_lanes16.hpp: 16 lanes per light curve, up to 256 resp. 512 rows).  The lists' lengths are known
// on the device only; the grid covers n_obj / 4 + 6 batches and the workgroups behind the last batch leave at once.  A light curve whose rows turn
// out not to ascend in time is appended to list `retry`.
#ifdef LCFE_DEBUG
constexpr double kLanesCanary = 0x1.5ca1ab1edeadp+900;
#endif
__global__ __launch_bounds__(64, 2) void stat_lanes_all_kernel(BatchView B, Bins bins, int retry, double* out, int ld, int col0) {
#ifdef LCFE_DEBUG
    // debug build: canary words around the buffers, verified when the workgroup's batch is done
    struct Framed { double pre[8]; StatLanesLds<32> L; double post[8]; };
    __shared__ Framed F;
    StatLanesLds<32>& L = F.L;
    if (threadIdx.x < 8) { F.pre[threadIdx.x] = kLanesCanary; F.post[threadIdx.x] = kLanesCanary; }
    __syncthreads();
    unsigned int lanes_bad[2] = {0u, 0u};
    struct CanaryCheck {
        Framed& F;
        unsigned int (&bad)[2];
        __device__ ~CanaryCheck() {
            __syncthreads();
            if (threadIdx.x < 8 && (F.pre[threadIdx.x] != kLanesCanary || F.post[threadIdx.x] != kLanesCanary)) atomicAdd(&g_lanes_check[1], 1u);
            if (bad[0]) { atomicAdd(&g_lanes_check[0], bad[0]); g_lanes_check[2] = bad[1]; }
        }
    } canary_check{F, lanes_bad};
    unsigned int* const bad_p = lanes_bad;
#else
    __shared__ StatLanesLds<32> L;
    unsigned int* const bad_p = nullptr;
#endif
    const LanesBuf lbuf = lanes_buf(L.buf, 64 * StatLanesLds<32>::STRIDE, bad_p), lall = lanes_buf(L.all_rows, 8 * 17, bad_p);
    const int c16 = bins.counts[kStatL16List], c32 = bins.counts[kStatL32List], c32x = bins.counts[kStatL32xList];
    const int cw16 = bins.counts[kStatW16List], cw32 = bins.counts[kStatW32List];
    const int nb16 = (c16 + 7) >> 3, nb32 = (c32 + 7) >> 3, nb32x = (c32x + 7) >> 3, nbw16 = (cw16 + 3) >> 2, nbw32 = (cw32 + 3) >> 2;
    int b = (int)blockIdx.x;
    int* rl = bins.lists + (int64_t)retry * bins.stride;
    int* rc = &bins.counts[retry];
    // (the long batches first: they are the ones whose tail would otherwise stick out)
    if (b < nbw32) {
        stat_lanes16_run<32, 32>(B.offsets, B.t, B.f, B.e, B.b, bins.lists + (int64_t)kStatW32List * bins.stride, cw32, b, lbuf,
                                 lall, out, ld, col0, rl, rc);
        return;
    }
    b -= nbw32;
    if (b < nbw16) {
        stat_lanes16_run<32, 16>(B.offsets, B.t, B.f, B.e, B.b, bins.lists + (int64_t)kStatW16List * bins.stride, cw16, b, lbuf,
                                 lall, out, ld, col0, rl, rc);
        return;
    }
    b -= nbw16;
    if (b < nb32x)
        stat_lanes_run<32, 32>(B.offsets, B.t, B.f, B.e, B.b, bins.lists + (int64_t)kStatL32xList * bins.stride, c32x, b, lbuf, lall, out, ld,
                               col0, rl, rc);
    else if (b < nb32x + nb32)
        stat_lanes_run<32, 16>(B.offsets, B.t, B.f, B.e, B.b, bins.lists + (int64_t)kStatL32List * bins.stride, c32, b - nb32x, lbuf, lall, out,
                               ld, col0, rl, rc);
    else if (b < nb32x + nb32 + nb16)
        stat_lanes_run<16, 16>(B.offsets, B.t, B.f, B.e, B.b, bins.lists + (int64_t)kStatL16List * bins.stride, c16, b - nb32x - nb32, lbuf,
                               lall, out, ld, col0, rl, rc);
}

// Which statistics kernel takes a light curve of up to 512 rows: ONE launch over the three tier lists (128 / 256 / 512
// rows), 8 lanes per light curve count its rows per band (1 byte per row is read, once) and the light curve is appended
// to the list of the cheapest lanes variant its bands fit:
//     rows <= 128   bands <= 16 rows (r, i: 32)   -> kStatL16List   (8 lanes per light curve, 16-row lanes)
//     rows <= 128   bands <= 32 (64)              -> kStatL32List   (8 lanes, 32-row lanes)
//     rows <= 256   bands <= 32 (64)              -> kStatL32xList  (8 lanes, 32-row lanes, 32 rows per lane loaded)
//     rows <= 256   bands <= 64 (128)             -> kStatW16List   (16 lanes per light curve)
//     rows <= 512   bands <= 64 (128)             -> kStatW32List   (16 lanes per light curve)
//     anything else (longer bands, unknown band codes, no rows)     -> kStatRetryList (one light curve per wavefront)
// The lists are private to the statistics set: the tier lists every other set reads are never appended to.  A workgroup
// takes 128 consecutive entries of one tier list (the long tier first); the lists' lengths are known on the device only,
// so the grid covers n_obj / 128 + 3 workgroups and the ones behind the last slice leave at once.  Lists are appended
// per workgroup with one atomic per destination.
constexpr int kPlanThreads = 1024;
constexpr int kPlanDst = 6;
template <int ITERS>
__device__ __forceinline__ unsigned long long stat_plan_count(const uint8_t* pb, int n, int j, bool& known) {
    // ITERS rows per lane, all loads in flight at once
    int bb[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int row = it * 8 + j;
        bb[it] = (row < n) ? (int)pb[row] : 256;
    }
    // rows per band: six 10-bit counters in one 64-bit word per lane (codes above 5 count in the bits that fall off)
    unsigned long long acc = 0;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int b = bb[it];
        known = known && (b < 6 || b == 256);
        acc += 1ull << (10 * ((b < 6) ? b : 6));
    }
    return acc;
}

__global__ __launch_bounds__(kPlanThreads) void stat_plan_kernel(BatchView B, Bins bins) {
    __shared__ int wcount[kPlanThreads / 64][kPlanDst];
    __shared__ int base[kPlanDst];
    constexpr int PER = kPlanThreads / 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 3, j = lane & 7;
    const int c0 = bins.counts[0], c1 = bins.counts[1], c2 = bins.counts[2];
    const int nb2 = (c2 + PER - 1) / PER, nb1 = (c1 + PER - 1) / PER, nb0 = (c0 + PER - 1) / PER;
    int blk = (int)blockIdx.x, tier, count;
    if (blk < nb2) { tier = 2; count = c2; }
    else if (blk < nb2 + nb1) { tier = 1; count = c1; blk -= nb2; }
    else if (blk < nb2 + nb1 + nb0) { tier = 0; count = c0; blk -= nb2 + nb1; }
    else return;
    const int64_t pos = (int64_t)blk * PER + wave * 8 + g;
    int obj = -1, n = 0;
    int64_t s0 = 0;
    if (pos < count) {
        obj = bins.lists[(int64_t)tier * bins.stride + pos];
        s0 = B.offsets[obj];
        n = (int)(B.offsets[obj + 1] - s0);
    }
    bool known = true;
    unsigned long long acc;
    if (tier == 0) acc = stat_plan_count<16>(B.b + s0, n, j, known);
    else if (tier == 1) acc = stat_plan_count<32>(B.b + s0, n, j, known);
    else acc = stat_plan_count<64>(B.b + s0, n, j, known);
    acc = GroupDev<8>::reduce((long long)acc, [](long long x, long long y) { return x + y; });
    int cnt[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) cnt[k] = (int)(acc >> (10 * k)) & 1023;
    known = GroupDev<8>::all(known);
    // what a lane of an 8-lane group must hold: the rows of u, g, z, y and half the rows of r, i
    int eff = cnt[0];
    eff = (cnt[1] > eff) ? cnt[1] : eff;
    eff = (cnt[4] > eff) ? cnt[4] : eff;
    eff = (cnt[5] > eff) ? cnt[5] : eff;
    eff = ((cnt[2] + 1) / 2 > eff) ? (cnt[2] + 1) / 2 : eff;
    eff = ((cnt[3] + 1) / 2 > eff) ? (cnt[3] + 1) / 2 : eff;
    const bool fits = known && n >= 1 && n <= 512;
    int cls = -1;
    if (obj >= 0 && j == 0) {
        cls = 5;
        if (fits && eff <= 64) {
            if (n <= 128) cls = (eff <= 16) ? 0 : ((eff <= 32) ? 1 : 3);
            else if (n <= 256) cls = (eff <= 32) ? 2 : 3;
            else cls = 4;
        }
    }
    int rank = 0;
#pragma unroll
    for (int c = 0; c < kPlanDst; ++c) {
        const unsigned long long m = __ballot(cls == c);
        if (cls == c) rank = WaveDev::prefix(m);
        if (lane == 0) wcount[wave][c] = popcll(m);
    }
    __syncthreads();
    auto dst_of = [](int c) { return (c == 0) ? kStatL16List : (c == 1) ? kStatL32List : (c == 2) ? kStatL32xList : (c == 3) ? kStatW16List : (c == 4) ? kStatW32List : kStatRetryList; };
    if (threadIdx.x < kPlanDst) {
        int total = 0;
        for (int w = 0; w < kPlanThreads / 64; ++w) {
            const int c = wcount[w][threadIdx.x];
            wcount[w][threadIdx.x] = total;
            total += c;
        }
        base[threadIdx.x] = total ? atomicAdd(&bins.counts[dst_of((int)threadIdx.x)], total) : 0;
    }
    __syncthreads();
    if (cls >= 0) bins.lists[(int64_t)dst_of(cls) * bins.stride + base[cls] + wcount[wave][cls] + rank] = obj;
}

// ---------------------------------------------------------------------------------------------------------------
// Bazin fits, fit by fit.  With one object per wavefront the six band fits of a light curve run side by side in
// 8-lane groups, but TRF's evaluation count is heavy-tailed (median 26, p99 556, cap 2000): the wave lasts as long as
// its slowest band and the other groups idle -- 37 % of the lane-group time did useful work.  Here the unit of work is
// ONE FIT: a partition pass writes every object's band-partitioned, time-sorted rows to global scratch and appends its
// fits to a list per band-length tier; the fit kernel runs a flat loop in which every 8-lane group advances the solver
// of its own fit by one phase per trip (trf.hpp: trf_outer / trf_inner) and takes the next fit from a device-side
// ticket counter the moment its own is finished -- the groups of a wave never wait for each other.  The arithmetic of
// a fit is exactly that of bazin_fit_band; only the schedule differs.
constexpr int kFitTiers = 4;
constexpr int kFitCaps[kFitTiers] = {32, 64, 128, 256};     // rows of one band
constexpr int kFitCountBase = 32;                           // counts[32 + t]: length of fit list t
constexpr int kFitTicketBase = 96;                          // tickets[96 + t]
struct FitWs {
    double* pt;            // band-partitioned copies of t / flux / err, indexed like the CSR arrays
    double* pf;
    double* pe;
    int* pboff;            // [n_obj][8] band offsets inside the object's slice
    int* fits;             // [kFitTiers][6 n_obj] fit ids (object * 8 + band)
    int64_t fit_stride;
};

// (times and fluxes are read from the partition pass's copy in global memory -- L2-resident, 16 bytes per row and
//  evaluation -- instead of a copy in LDS: 2.5 KB per fit at 32 rows instead of 3 KB = eight wavefronts per CU where six fitted)
template <int BCAP>
struct FitRegion {
    double r[BCAP], rn[BCAP], w[BCAP];     // rn doubles as the key scratch of the median before the first trial point
    double A[6][BCAP + 5];
    double slot[2];
};

// partition pass: one object per wavefront (tier lists of the feature sets), up to `chunk` objects per ticket
template <int CAP>
__global__ __launch_bounds__(64, 2) void bazin_partition_kernel(BatchView B, Bins bins, int bin, int nan_from, FitWs F, double* out, int ld,
                                                                int col0, int32_t* status, int st_ld, int st0,
                                                                unsigned long long* ticket, int chunk) {
    __shared__ ObjLds<CAP> L;
    __shared__ long long next_ticket;
    __shared__ int loc[kFitTiers][64], nloc[kFitTiers], base_of[kFitTiers];
    using W = WaveDev;
    const int count = bins.counts[bin];
    const int* list = bins.lists + (int64_t)bin * bins.stride;
    const int per_wave = count / (4 * (int)gridDim.x);
    chunk = (per_wave < 1) ? 1 : ((per_wave < chunk) ? per_wave : chunk);
    for (;;) {
        if (threadIdx.x == 0) { next_ticket = (long long)atomicAdd(ticket, 1ull); for (int q = 0; q < kFitTiers; ++q) nloc[q] = 0; }
        __syncthreads();
        const int64_t base = next_ticket * chunk;
        __syncthreads();
        if (base >= count) break;
        const int nk = (count - base < chunk) ? (int)(count - base) : chunk;
        for (int k = 0; k < nk; ++k) {
            const int64_t i = list[base + k];
            const int64_t s = B.offsets[i];
            const int n = (int)(B.offsets[i + 1] - s);
            ObjIn in{B.t + s, B.f + s, B.e + s, B.b + s, n, qnan()};
            stage_object<W, CAP>(in, L);
            const int nb = L.boff[6];
            for (int q = threadIdx.x; q < nb; q += 64) { F.pt[s + q] = L.bt[q]; F.pf[s + q] = L.bf[q]; F.pe[s + q] = L.be[q]; }
            if (threadIdx.x < 8) F.pboff[i * 8 + threadIdx.x] = L.boff[threadIdx.x];
            int mb = 0;
#pragma unroll
            for (int b = 0; b < 6; ++b) { const int m = L.boff[b + 1] - L.boff[b]; mb = (m > mb) ? m : mb; }
            if (mb > kFitCaps[kFitTiers - 1]) {
                // a band beyond the largest fit tier: the whole object goes to the object-level kernel, which takes
                // light curves of up to 1024 rows; beyond that the fit is not available (NaN, status -100)
                if (n <= 1024) {
                    if (threadIdx.x == 0) {
                        const int slot = atomicAdd(&bins.counts[kBazinFallbackList], 1);
                        bins.lists[(int64_t)kBazinFallbackList * bins.stride + slot] = (int)i;
                    }
                } else {
                    // the long-object tier takes it (when the caller's workspace has its slabs); NaN and -100 until then
                    fill_row_nan<W>(out + i * (int64_t)ld + col0, BAZIN_NCOL);
                    if (status && threadIdx.x < 12) status[i * (int64_t)st_ld + st0 + threadIdx.x] = -100;
                    if (threadIdx.x == 0) {
                        const int slot = atomicAdd(&bins.counts[kBazinLongList], 1);
                        bins.lists[(int64_t)kBazinLongList * bins.stride + slot] = (int)i;
                    }
                }
            } else if (threadIdx.x < 6) {
                const int b = threadIdx.x;
                const int m = L.boff[b + 1] - L.boff[b];
                if (m < 5) {                                                    // bazin_fitting.py:76-87
                    double* o8 = out + i * (int64_t)ld + col0 + 8 * b;
                    for (int q = 0; q < 8; ++q) o8[q] = qnan();
                    if (status) { status[i * (int64_t)st_ld + st0 + 2 * b] = TRF_FAIL_TOO_FEW; status[i * (int64_t)st_ld + st0 + 2 * b + 1] = 0; }
                } else {
                    const int tier = (m <= kFitCaps[0]) ? 0 : ((m <= kFitCaps[1]) ? 1 : ((m <= kFitCaps[2]) ? 2 : 3));
                    const int slot = atomicAdd(&nloc[tier], 1);                 // LDS atomic: order inside a chunk is free
                    loc[tier][slot] = (int)i * 8 + b;
                }
            }
            __syncthreads();
        }
        // one global atomic per tier and chunk
        if (threadIdx.x < kFitTiers) base_of[threadIdx.x] = nloc[threadIdx.x] ? atomicAdd(&bins.counts[kFitCountBase + threadIdx.x], nloc[threadIdx.x]) : 0;
        __syncthreads();
        for (int tr = 0; tr < kFitTiers; ++tr)
            for (int q = threadIdx.x; q < nloc[tr]; q += 64) F.fits[tr * F.fit_stride + base_of[tr] + q] = loc[tr][q];
        __syncthreads();
    }
    nan_fill_bins<W>(bins, 0, nan_from, out, ld, col0, BAZIN_NCOL, status, st_ld, st0, 12);
}

enum { FIT_IDLE = 3, FIT_EXIT = 4 };      // beside TRF_PH_OUTER / INNER / DONE

template <int BCAP> struct fit_waves { static constexpr int N = (BCAP <= 32) ? 2 : 1; };
// lane groups of a wave that take fits: all eight while their LDS regions fit next to each other, six in the 256-row tier
template <int BCAP> struct fit_groups { static constexpr int N = (BCAP <= 128) ? 8 : 6; };

template <int BCAP>
__global__ __launch_bounds__(64, (fit_waves<BCAP>::N)) void bazin_fit_kernel(BatchView B, Bins bins, FitWs F, int tier, double* out, int ld, int col0,
                                                            int32_t* status, int st_ld, int st0, unsigned long long* ticket) {
    using G = GroupDev<8>;
    constexpr int NG = fit_groups<BCAP>::N;
    __shared__ FitRegion<BCAP> R[NG];
    FitRegion<BCAP>& Rg = R[(G::group_id() < NG) ? G::group_id() : 0];
    TrfView<5> V;
#pragma unroll
    for (int k = 0; k < 6; ++k) V.A[k] = Rg.A[k];
    V.r = Rg.r; V.rn = Rg.rn; V.w = Rg.w;
    const int count = bins.counts[kFitCountBase + tier];
    const int* list = F.fits + (int64_t)tier * F.fit_stride;
    const int gl = G::lane();
    const int leader = (int)(threadIdx.x & 63) & ~7;
    TrfState<5> Z;
    Z.phase = (G::group_id() < NG) ? FIT_IDLE : FIT_EXIT;
    int m = 0;
    int64_t obj = 0, src = 0;
    int band = 0;
    for (;;) {
        if (Z.phase == FIT_IDLE) {
            int tk = 0;
            if (gl == 0) tk = (int)atomicAdd(ticket, 1ull);
            tk = __shfl(tk, leader, 64);
            if (tk >= count) Z.phase = FIT_EXIT;
            else {
                const int id = list[tk];
                obj = id >> 3;
                band = id & 7;
                const int b0 = F.pboff[obj * 8 + band];
                m = F.pboff[obj * 8 + band + 1] - b0;
                src = B.offsets[obj] + b0;
                bazin_prepare<G, TrfView<5>>(F.pt + src, F.pf + src, F.pe + src, m, V, Rg.slot, reinterpret_cast<unsigned long long*>(Rg.rn), Z);
                trf_begin<G, BazinModel, TrfView<5>>(BazinModel(), F.pt + src, F.pf + src, m, Z, V);
            }
        }
        if (__ballot(Z.phase != FIT_EXIT) == 0ull) break;
        if (Z.phase == TRF_PH_OUTER) trf_outer<G, BazinModel, TrfView<5>>(m, Z, V);
        if (Z.phase == TRF_PH_INNER) trf_inner<G, BazinModel, TrfView<5>>(BazinModel(), F.pt + src, F.pf + src, m, Z, V);
        if (Z.phase == TRF_PH_DONE) {
            bazin_finish<G>(F.pt + src, F.pf + src, F.pe + src, m, Z, out + obj * (int64_t)ld + col0 + 8 * band);
            if (status && gl == 0) {
                status[obj * (int64_t)st_ld + st0 + 2 * band] = Z.res.status;
                status[obj * (int64_t)st_ld + st0 + 2 * band + 1] = Z.res.nfev;
            }
            G::sync();
            Z.phase = FIT_IDLE;
        }
    }
}

// the four cross-band columns of every object (bazin_fitting.py:217-249) once all its fits are in
__global__ __launch_bounds__(256) void bazin_cross_kernel(BatchView B, double* out, int ld, int col0) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B.n_obj) return;
    double o[BAZIN_NCOL];
    double* row = out + i * (int64_t)ld + col0;
    for (int k = 0; k < 48; ++k) o[k] = row[k];
    bazin_cross_band(o);
    for (int k = 48; k < 52; ++k) row[k] = o[k];
}

// ---------------------------------------------------------------------------------------------------------------
// Decline-model fits (v55 power-law set), fit by fit: same scheme as the Bazin fits above.  A partition pass writes,
// per object and band g/r/i, the post-peak rows (times relative to the peak, fluxes) and their peak flux and total
// sum of squares, and queues the band's nine fits: the seven power laws (two parameters, the exponent is data) in
// list A, the exponential and the linear model (three parameters) in list B, per tier of the post-peak row count.
constexpr int kPlCountA = 40, kPlCountB = 44;               // counts[40 + t], counts[44 + t]
constexpr int kPlTicketA = 104, kPlTicketB = 108;           // tickets[...]
struct PlWs {
    double* tp;            // post-peak times of band j of object i at offsets[i] + boff[j + 1] ..., indexed like the CSR arrays
    double* fp;
    double* peak;          // [n_obj][3]
    double* sstot;         // [n_obj][3]
    int* pboff;            // [n_obj][8]
    int* kk;               // [n_obj][3] post-peak rows (-1: no fits)
    int* fitsA;            // [kFitTiers][21 n_obj]   ids: object * 32 + band * 9 + model
    int* fitsB;            // [kFitTiers][6 n_obj]
    int64_t strideA, strideB;
};

// (the decline fits keep their rows in LDS: read from global memory their 32-row tiers ran 3-5 % slower -- LDS is not
//  what limits their occupancy)
template <int N, int BCAP>
struct PlRegion {
    double t[BCAP], f[BCAP];
    double r[BCAP], rn[BCAP], w[BCAP];
    double A[N + 1][BCAP + N];
};

template <int CAP>
__global__ __launch_bounds__(64, 2) void powerlaw_partition_kernel(BatchView B, Bins bins, int bin, int nan_from, PlWs F, double* out, int ld,
                                                                   int col0, int32_t* status, int st_ld, int st0,
                                                                   unsigned long long* ticket, int chunk) {
    __shared__ ObjLds<CAP> L;
    __shared__ long long next_ticket;
    using W = WaveDev;
    const int count = bins.counts[bin];
    const int* list = bins.lists + (int64_t)bin * bins.stride;
    const int per_wave = count / (4 * (int)gridDim.x);
    chunk = (per_wave < 1) ? 1 : ((per_wave < chunk) ? per_wave : chunk);
    for (;;) {
        if (threadIdx.x == 0) next_ticket = (long long)atomicAdd(ticket, 1ull);
        __syncthreads();
        const int64_t base = next_ticket * chunk;
        __syncthreads();
        if (base >= count) break;
        const int nk = (count - base < chunk) ? (int)(count - base) : chunk;
        for (int q = 0; q < nk; ++q) {
            const int64_t i = list[base + q];
            const int64_t s = B.offsets[i];
            const int n = (int)(B.offsets[i + 1] - s);
            ObjIn in{B.t + s, B.f + s, B.e + s, B.b + s, n, qnan()};
            stage_object<W, CAP>(in, L);
            if (threadIdx.x < 8) F.pboff[i * 8 + threadIdx.x] = L.boff[threadIdx.x];
            int kb[3];
            for (int j = 0; j < 3; ++j) {
                const int b0 = L.boff[j + 1], m = L.boff[j + 2] - b0;
                int k, first;
                double peak_flux, ss_tot;
                powerlaw_band_prepare<W>(L.bt + b0, L.bf + b0, m, F.tp + s + b0, F.fp + s + b0, k, first, peak_flux, ss_tot);
                kb[j] = k;
                if (threadIdx.x == 0) { F.kk[i * 3 + j] = k; F.peak[i * 3 + j] = peak_flux; F.sstot[i * 3 + j] = ss_tot; }
            }
            const int kmax = (kb[0] > kb[1]) ? ((kb[0] > kb[2]) ? kb[0] : kb[2]) : ((kb[1] > kb[2]) ? kb[1] : kb[2]);
            if (kmax > kFitCaps[kFitTiers - 1]) {
                // more post-peak rows than the largest fit tier: object-level kernel (up to 1024 rows), else not available
                if (n <= 1024) {
                    if (threadIdx.x == 0) {
                        const int slot = atomicAdd(&bins.counts[kPowerlawFallbackList], 1);
                        bins.lists[(int64_t)kPowerlawFallbackList * bins.stride + slot] = (int)i;
                    }
                } else {
                    fill_row_nan<W>(out + i * (int64_t)ld + col0, POWERLAW_NCOL);
                    if (status && threadIdx.x < 54) status[i * (int64_t)st_ld + st0 + threadIdx.x] = -100;
                    if (threadIdx.x == 0) {
                        const int slot = atomicAdd(&bins.counts[kPowerlawLongList], 1);
                        bins.lists[(int64_t)kPowerlawLongList * bins.stride + slot] = (int)i;
                    }
                }
            } else if (threadIdx.x < 3) {
                const int j = threadIdx.x;
                const int k = kb[j];
                if (k < 0) {                                                    // train_v55_powerlaw.py:150-151,162-163
                    double* o9 = out + i * (int64_t)ld + col0 + 9 * j;
                    for (int id = 0; id < 9; ++id) {
                        o9[id] = qnan();
                        if (status) { status[i * (int64_t)st_ld + st0 + 18 * j + 2 * id] = TRF_FAIL_TOO_FEW; status[i * (int64_t)st_ld + st0 + 18 * j + 2 * id + 1] = 0; }
                    }
                } else {
                    const int tier = (k <= kFitCaps[0]) ? 0 : ((k <= kFitCaps[1]) ? 1 : ((k <= kFitCaps[2]) ? 2 : 3));
                    const int a0 = atomicAdd(&bins.counts[kPlCountA + tier], 7);
                    for (int id = 0; id < 7; ++id) F.fitsA[tier * F.strideA + a0 + id] = (int)i * 32 + j * 9 + id;
                    const int b0 = atomicAdd(&bins.counts[kPlCountB + tier], 2);
                    F.fitsB[tier * F.strideB + b0] = (int)i * 32 + j * 9 + 7;
                    F.fitsB[tier * F.strideB + b0 + 1] = (int)i * 32 + j * 9 + 8;
                }
            }
            __syncthreads();
        }
    }
    nan_fill_bins<W>(bins, 0, nan_from, out, ld, col0, POWERLAW_NCOL, status, st_ld, st0, 54);
}

// N = 2: the seven power laws (PowerModel, exponent from the fit id); N = 3: exponential and linear (Decline3Model)
template <int N> struct pl_model;
template <> struct pl_model<2> { using type = PowerModel; static __device__ __forceinline__ PowerModel make(int id) { return PowerModel{decline_exponent(id)}; } };
template <> struct pl_model<3> { using type = Decline3Model; static __device__ __forceinline__ Decline3Model make(int id) { return Decline3Model{id}; } };

template <int N, int BCAP>
__global__ __launch_bounds__(64, (BCAP <= 64 ? 2 : 1)) void powerlaw_fit_kernel(BatchView B, Bins bins, PlWs F, int tier, double* out, int ld,
                                                                                  int col0, int32_t* status, int st_ld, int st0,
                                                                                  unsigned long long* ticket) {
    using G = GroupDev<8>;
    using M = typename pl_model<N>::type;
    constexpr int NG = fit_groups<BCAP>::N;
    __shared__ PlRegion<N, BCAP> R[NG];
    PlRegion<N, BCAP>& Rg = R[(G::group_id() < NG) ? G::group_id() : 0];
    TrfView<N> V;
#pragma unroll
    for (int k = 0; k <= N; ++k) V.A[k] = Rg.A[k];
    V.r = Rg.r; V.rn = Rg.rn; V.w = Rg.w;
    const int count = bins.counts[((N == 2) ? kPlCountA : kPlCountB) + tier];
    const int* list = (N == 2) ? F.fitsA + (int64_t)tier * F.strideA : F.fitsB + (int64_t)tier * F.strideB;
    const int gl = G::lane();
    const int leader = (int)(threadIdx.x & 63) & ~7;
    TrfState<N> Z;
    Z.phase = (G::group_id() < NG) ? FIT_IDLE : FIT_EXIT;
    M model = pl_model<N>::make(N == 2 ? 0 : 7);
    int k = 0, id = 0, j = 0;
    int64_t obj = 0;
    double ss_tot = 0;
    for (;;) {
        if (Z.phase == FIT_IDLE) {
            int tk = 0;
            if (gl == 0) tk = (int)atomicAdd(ticket, 1ull);
            tk = __shfl(tk, leader, 64);
            if (tk >= count) Z.phase = FIT_EXIT;
            else {
                const int fid = list[tk];
                obj = fid >> 5;
                j = (fid & 31) / 9;
                id = (fid & 31) - 9 * j;
                k = F.kk[obj * 3 + j];
                ss_tot = F.sstot[obj * 3 + j];
                const int64_t src = B.offsets[obj] + F.pboff[obj * 8 + j + 1];
                for (int i = gl; i < k; i += 8) { Rg.t[i] = F.tp[src + i]; Rg.f[i] = F.fp[src + i]; Rg.w[i] = 1.0; }
                model = pl_model<N>::make(id);
                decline_setup<N>(id, F.peak[obj * 3 + j], Z);
                G::sync();
                trf_begin<G, M, TrfView<N>>(model, Rg.t, Rg.f, k, Z, V);
            }
        }
        if (__ballot(Z.phase != FIT_EXIT) == 0ull) break;
        if (Z.phase == TRF_PH_OUTER) trf_outer<G, M, TrfView<N>>(k, Z, V);
        if (Z.phase == TRF_PH_INNER) trf_inner<G, M, TrfView<N>>(model, Rg.t, Rg.f, k, Z, V);
        if (Z.phase == TRF_PH_DONE) {
            decline_finish<G, M>(model, Rg.t, Rg.f, k, ss_tot, Z, out + obj * (int64_t)ld + col0 + 9 * j + id);
            if (status && gl == 0) {
                status[obj * (int64_t)st_ld + st0 + 18 * j + 2 * id] = Z.res.status;
                status[obj * (int64_t)st_ld + st0 + 18 * j + 2 * id + 1] = Z.res.nfev;
            }
            G::sync();
            Z.phase = FIT_IDLE;
        }
    }
}

int num_cus(int dev);

// ---- 2-D GP: one light curve per workgroup (256/512/1024 threads by tier); Gram matrix as 16x16
// lower-triangle tiles in LDS (NP <= 160) or, for longer light curves, in a per-workgroup slab of
// global scratch.  Objects come from the tier's index list through a ticket counter (one object per
// ticket: an L-BFGS-B run is 10^5..10^7 cycles and heavy-tailed).
constexpr int kGpGlobalGrid = 256;
constexpr int kGpSmallGrid = 512;
constexpr size_t kGpSmallBytes = (size_t)kGpSmallGrid * gp_store_doubles(kGpSmallNP) * 8;
constexpr size_t kGpMidBytes = (size_t)kGpGlobalGrid * gp_store_doubles(kGpMidNP) * 8;
constexpr size_t kGpGlobalBytes = (size_t)kGpGlobalGrid * gp_store_doubles(kGpGlobalNP) * 8;
// the 112- and 160-row tiers can keep their matrix in global scratch as well (L2-resident slabs): the LDS then holds
// only the two pivot panels, two to four workgroups share a CU and the sweep takes two pivot tiles per pass.  Measured
// (125,000 objects): serialised, both tiers gain 8-9 %; with the fit kernels running beside the GP (the default
// schedule) the 112-row tier in global scratch gains 1.5 % end to end and the 160-row tier LOSES 5 % -- so only the former.
constexpr bool kGp112Global = true, kGp160Global = false;
constexpr int kGp112Grid = 1024, kGp160Grid = 512;
constexpr size_t kGp112Bytes = kGp112Global ? (size_t)kGp112Grid * gp_store_doubles(112) * 8 : 0;
constexpr size_t kGp160Bytes = kGp160Global ? (size_t)kGp160Grid * gp_store_doubles(160) * 8 : 0;
constexpr size_t kGpScratchBytes = kGpSmallBytes + kGpMidBytes + kGpGlobalBytes + kGp112Bytes + kGp160Bytes;

template <int NP> struct gp_threads { static constexpr int T = (NP >= 768) ? 1024 : ((NP >= 160) ? 512 : 256); };
// the two global-scratch tiers whose second pivot panel fits LDS sweep two pivot tiles per pass over the matrix
template <int NP, bool GLOBAL_K> struct gp_fuse { static constexpr bool F = GLOBAL_K && NP <= 512; };
template <int NP> struct gp_grid_cap { static constexpr int G = (NP == 112) ? kGp112Grid : ((NP == 160) ? kGp160Grid : ((NP == kGpSmallNP) ? kGpSmallGrid : kGpGlobalGrid)); };

// waves per SIMD to leave room for: the 64- and 112-row tiers fit two or more workgroups per CU in LDS,
// so their register budget is halved (the L-BFGS-B driver spills a little, the sweep does not)
template <int NP, bool GLOBAL_K> struct gp_waves { static constexpr int N = (NP <= 64) ? 3 : (GLOBAL_K ? ((NP == 512) ? 2 : 4) : 2); };

// the long-object tier of the 2-D GP: light curves of more than 767 rows, up to kGpLongNP - 1 valid points; Gram matrix AND
// working set (panels, point arrays, optimiser state) in per-workgroup slabs of global scratch
constexpr int kGpLongNP = 2048;
constexpr int kGpLongGrid = 8;
// (no pivot look-ahead: its staging tile is handed over between lanes of one wavefront under wave-level fences, which the
//  tiers rely on for LDS only)
template <int NP> struct gp_working_set { using type = GpLds<NP, gp_threads<NP>::T / 64, false, false, false>; };
constexpr size_t kGpLongSBytes = (sizeof(gp_working_set<kGpLongNP>::type) + 255) & ~(size_t)255;
constexpr size_t kGpLongBytes = (size_t)kGpLongGrid * ((size_t)gp_store_doubles(kGpLongNP) * 8 + kGpLongSBytes);

template <int NP, bool GLOBAL_K>
__global__ __launch_bounds__(gp_threads<NP>::T, (gp_waves<NP, GLOBAL_K>::N)) void gp_kernel(BatchView B, Bins bins, int bin, int nan_from, double* out, int ld,
                                                 int col0, int32_t* status, int st_ld, int st0, double* kscratch,
                                                 unsigned long long* ticket) {
    using W = BlockDev<gp_threads<NP>::T>;
    __shared__ GpLds<NP, W::NWAVES, gp_fuse<NP, GLOBAL_K>::F> S;
    __shared__ double Klds[GLOBAL_K ? 1 : gp_store_doubles(NP)];
    __shared__ long long next_ticket;
    double* Kg = kscratch + (size_t)blockIdx.x * (size_t)gp_store_doubles(NP);
    const int count = bins.counts[kNumBins + bin];
    const int* list = bins.lists + (int64_t)(kGpSortedList + bin) * bins.stride;      // longest first (gp_sort_kernel)
    for (;;) {
        if (threadIdx.x == 0) next_ticket = (long long)atomicAdd(ticket, 1ull);
        __syncthreads();
        const int64_t pos = next_ticket;
        __syncthreads();
        if (pos >= count) break;
        const int64_t i = list[pos];
        const int64_t s = B.offsets[i];
        const int64_t n64 = B.offsets[i + 1] - s;
        double* row = out + i * (int64_t)ld + col0;
        int32_t* st = status ? status + i * (int64_t)st_ld + st0 : nullptr;
        ObjIn in{B.t + s, B.f + s, B.e + s, B.b + s, (int)n64, qnan()};
        if constexpr (GLOBAL_K) {
            gp_object<W, NP>(in, S, [&](const double* x, int n, double& f, double* g, bool need) {
                gp_eval<W, NP, global_double*>(x, n, S, (global_double*)Kg, f, g, need); }, st);
        } else {
            gp_object<W, NP>(in, S, [&](const double* x, int n, double& f, double* g, bool need) {
                gp_eval<W, NP, lds_double*>(x, n, S, (lds_double*)Klds, f, g, need); }, st);
        }
        store_row<W>(S.out, row, GP_NCOL);
        __syncthreads();
    }
    nan_fill_bins<W>(bins, 1, nan_from, out, ld, col0, GP_NCOL, status, st_ld, st0, 4);
}

// the same loop for the long-object tier (slabs: kGpLongGrid Gram matrices, then kGpLongGrid working sets)
__global__ __launch_bounds__(gp_threads<kGpLongNP>::T, 1) void gp_long_kernel(BatchView B, Bins bins, int bin, double* out, int ld, int col0,
                                                                             int32_t* status, int st_ld, int st0, char* slabs,
                                                                             unsigned long long* ticket) {
    constexpr int NP = kGpLongNP;
    using W = BlockDev<gp_threads<NP>::T>;
    using SL = gp_working_set<NP>::type;
    __shared__ long long next_ticket;
    double* Kg = reinterpret_cast<double*>(slabs) + (size_t)blockIdx.x * (size_t)gp_store_doubles(NP);
    SL& S = *reinterpret_cast<SL*>(slabs + (size_t)kGpLongGrid * gp_store_doubles(NP) * 8 + (size_t)blockIdx.x * kGpLongSBytes);
    const int count = bins.counts[kNumBins + bin];
    const int* list = bins.lists + (int64_t)(kNumBins + bin) * bins.stride;
    for (;;) {
        if (threadIdx.x == 0) next_ticket = (long long)atomicAdd(ticket, 1ull);
        __syncthreads();
        const int64_t pos = next_ticket;
        __syncthreads();
        if (pos >= count) break;
        const int64_t i = list[pos];
        const int64_t s = B.offsets[i];
        const int64_t n64 = B.offsets[i + 1] - s;
        int32_t* st = status ? status + i * (int64_t)st_ld + st0 : nullptr;
        ObjIn in{B.t + s, B.f + s, B.e + s, B.b + s, (int)n64, qnan()};
        gp_object<W, NP>(in, S, [&](const double* x, int n, double& f, double* g, bool need) {
            gp_eval<W, NP, global_double*>(x, n, S, (global_double*)Kg, f, g, need); }, st);
        store_row<W>(S.out, out + i * (int64_t)ld + col0, GP_NCOL);
        __syncthreads();
    }
}

// The tickets of a GP tier are served longest light curve first: an evaluation costs ~N^3 and a run 10-100 evaluations, so
// in file order the last tickets of a tier can be its most expensive light curves (a 500-row one is 60 ms on ONE workgroup
// while the rest of the grid has drained; simulated on the bench shard, the 512-row tier ends 14 % after the ideal in file
// order and 2 % after it longest first).  One workgroup per tier: counting sort of the tier's list by row count into the
// tier's second list.  The order inside a row count is whatever the atomics give -- results do not depend on the order.
constexpr int kGpSortThreads = 1024;
__global__ __launch_bounds__(kGpSortThreads) void gp_sort_kernel(const int64_t* offsets, Bins bins) {
    const int tier = blockIdx.x;
    const int hi = (tier == 0) ? 63 : (tier == 1) ? 111 : (tier == 2) ? 159 : (tier == 3) ? kGpSmallNP - 1 : (tier == 4) ? kGpMidNP - 1 : kGpGlobalNP - 1;
    const int count = bins.counts[kNumBins + tier];
    const int* src = bins.lists + (int64_t)(kNumBins + tier) * bins.stride;
    int* dst = bins.lists + (int64_t)(kGpSortedList + tier) * bins.stride;
    __shared__ int hist[1024], wsum[kGpSortThreads / 64];
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < count; k += kGpSortThreads) {
        const int i = src[k];
        const int key = hi - (int)(offsets[i + 1] - offsets[i]);             // 0 = the longest light curve the tier takes
        atomicAdd(&hist[(key < 0) ? 0 : ((key > 1023) ? 1023 : key)], 1);
    }
    __syncthreads();
    // exclusive scan of the 1024 counters: inside the wavefronts, then over the 16 wavefront totals
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mine = hist[threadIdx.x];
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    __syncthreads();
    hist[threadIdx.x] = before + incl - mine;
    __syncthreads();
    for (int k = threadIdx.x; k < count; k += kGpSortThreads) {
        const int i = src[k];
        const int key = hi - (int)(offsets[i + 1] - offsets[i]);
        const int at = atomicAdd(&hist[(key < 0) ? 0 : ((key > 1023) ? 1023 : key)], 1);
        dst[at] = i;
    }
}

template <int NP, bool GLOBAL_K>
int launch_gp_tier(const BatchView& B, const Bins& bins, int bin, int nan_from, double* out, int ld, int col0,
                   int32_t* status, int st_ld, int st0, hipStream_t stream, int dev, double* kscratch,
                   unsigned long long* ticket, int64_t grid_cap = 0) {
    int per_cu = 0;
    constexpr int threads = gp_threads<NP>::T;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gp_kernel<NP, GLOBAL_K>, threads, 0));
    if (per_cu < 1) per_cu = 1;
    int64_t grid = (int64_t)num_cus(dev) * per_cu;
    if (GLOBAL_K && grid > gp_grid_cap<NP>::G) grid = gp_grid_cap<NP>::G;
    if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
    {
        // tuning knob: LCFE_GP_GRID_<rows>=k caps the workgroups of a tier (fewer Gram matrices in flight = more of them in L2)
        char name[40];
        snprintf(name, sizeof name, "LCFE_GP_GRID_%d", NP);
        const char* e = getenv(name);
        if (e && atoi(e) > 0 && atoi(e) < grid) grid = atoi(e);
    }
    if (grid > B.n_obj) grid = B.n_obj;
    if (grid < 1) return 0;
    hipLaunchKernelGGL((gp_kernel<NP, GLOBAL_K>), dim3((unsigned)grid), dim3(threads), 0, stream, B, bins, bin, nan_from,
                       out, ld, col0, status, st_ld, st0, kscratch, ticket);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_gp(const BatchView& B, const Bins& bins, int64_t max_len, double* out, int ld, int col0, int32_t* status,
              int st_ld, int st0, const hipStream_t* gs, int ngs, int dev, double* kscratch,
              size_t kscratch_bytes, int* n_launch, unsigned long long* tickets, char* long_slabs) {
    hipStream_t stream = gs[0], stream2 = gs[(ngs > 1) ? 1 : 0];
    // (A variant that keeps the matrix in the REGISTERS of the workgroup -- 2-D block-cyclic tiles,
    // register-tiled outer products -- was built and measured: slower on every tier, because hipcc
    // spends 412-512 registers per lane on the unrolled tile passes and spills at 1024 threads.)
    const int caps[6] = {63, 111, 159, kGpSmallNP - 1, kGpMidNP - 1, kGpGlobalNP - 1};
    int last = 0;
    while (last < 5 && caps[last] < max_len) ++last;
    if (kscratch_bytes < kGpScratchBytes)
        return fail_msg("lcfe_extract_device: workspace too small for the GP global tiers");
    double* k_small = kscratch;
    double* k_mid = kscratch + kGpSmallBytes / 8;
    double* k_glob = k_mid + kGpMidBytes / 8;
    double* k_112 = k_glob + kGpGlobalBytes / 8;
    double* k_160 = k_112 + kGp112Bytes / 8;
    // every tier's list longest light curve first (on the first GP stream; the others wait for it)
    {
        hipLaunchKernelGGL(gp_sort_kernel, dim3((unsigned)(last + 1)), dim3(kGpSortThreads), 0, gs[0], B.offsets, bins);
        HIP_TRY(hipGetLastError());
        ++*n_launch;
        if (ngs > 1) {
            hipEvent_t sorted;
            HIP_TRY(hipEventCreateWithFlags(&sorted, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(sorted, gs[0]));
            for (int k = 1; k < ngs; ++k) HIP_TRY(hipStreamWaitEvent(gs[k], sorted, 0));
            (void)hipEventDestroy(sorted);
        }
    }
    // Launch plan: which stream takes which tiers, in which order, and on how many workgroups at most.  Default (two GP
    // streams): longest tier first, tiers alternating between the streams -- the heavy-tailed end of one tier (single objects
    // of up to 50 ms) overlaps with the start of the next.  The 512-row tier holds a whole CU per workgroup (160 KB of LDS)
    // for the first half second of the step and is bound by the MFMA pipe: when the fit kernels run beside the GP (side
    // streams), 192 of the 256 CUs are enough for it and the other 64 let the fits overlap from the start -- +4.5 % light
    // curves/s for the whole step (75.2 k against 72.0 k, two runs each), where the GP on its own would lose 12 %.
    // LCFE_GP_PLAN overrides it for experiments: streams separated by '|', tiers (0 = 64 rows .. 5 = 768 rows) by ',', an
    // optional ':cap' after a tier, e.g. "4:192,2,0|3,1" (the default with two streams).
    struct PlanItem { int tier, stream; int64_t cap; };
    PlanItem plan[12];
    int n_plan = 0;
    {
        static const char* env = getenv("LCFE_GP_PLAN");
        bool seen[6] = {false, false, false, false, false, false};
        if (env && ngs > 1) {
            int st = 0;
            for (const char* c = env; *c && n_plan < 12;) {
                if (*c == '|') { ++st; ++c; continue; }
                if (*c == ',') { ++c; continue; }
                if (*c < '0' || *c > '5') return fail_msg("lcfe_extract_device: malformed LCFE_GP_PLAN");
                PlanItem it{*c - '0', (st < ngs) ? st : ngs - 1, 0};
                ++c;
                if (*c == ':') { it.cap = atoi(c + 1); ++c; while (*c >= '0' && *c <= '9') ++c; }
                if (it.tier <= last && !seen[it.tier]) { seen[it.tier] = true; plan[n_plan++] = it; }
            }
        }
        for (int ti = last, pos = 0; ti >= 0; --ti, ++pos) {
            if (env && ngs > 1) { if (!seen[ti]) plan[n_plan++] = PlanItem{ti, 0, 0}; continue; }
            plan[n_plan++] = PlanItem{ti, pos % ngs, (ti == 4 && stream2 != stream) ? 192 : 0};
        }
    }
    for (int pi = 0; pi < n_plan; ++pi) {
        const int ti = plan[pi].tier;
        const int64_t cap = plan[pi].cap;
        const int nan_from = (ti == last) ? ti + 1 : kNumBins;
        unsigned long long* tk = tickets + SET_GP2D * 8 + ti;
        hipStream_t q = gs[plan[pi].stream];
        int rc = 0;
        switch (ti) {
            case 0: rc = launch_gp_tier<64, false>(B, bins, ti, nan_from, out, ld, col0, status, st_ld, st0, q, dev, nullptr, tk, cap); break;
            case 1: rc = launch_gp_tier<112, kGp112Global>(B, bins, ti, nan_from, out, ld, col0, status, st_ld, st0, q, dev, k_112, tk, cap); break;
            case 2: rc = launch_gp_tier<160, kGp160Global>(B, bins, ti, nan_from, out, ld, col0, status, st_ld, st0, q, dev, k_160, tk, cap); break;
            case 3: rc = launch_gp_tier<kGpSmallNP, true>(B, bins, ti, nan_from, out, ld, col0, status, st_ld, st0, q, dev, k_small, tk, cap); break;
            case 4: rc = launch_gp_tier<kGpMidNP, true>(B, bins, ti, nan_from, out, ld, col0, status, st_ld, st0, q, dev, k_mid, tk, cap); break;
            case 5: rc = launch_gp_tier<kGpGlobalNP, true>(B, bins, ti, nan_from, out, ld, col0, status, st_ld, st0, q, dev, k_glob, tk, cap); break;
        }
        if (rc) return rc;
        ++*n_launch;
        // light curves of more than 767 rows (bin 6; NaN rows and status -100 from the launch above until this one has run)
        if (ti == last && last == 5 && long_slabs) {
            hipLaunchKernelGGL(gp_long_kernel, dim3(kGpLongGrid), dim3(gp_threads<kGpLongNP>::T), 0, q, B, bins, 6, out, ld, col0, status,
                               st_ld, st0, long_slabs, tickets + SET_GP2D * 8 + 7);
            HIP_TRY(hipGetLastError());
            ++*n_launch;
        }
    }
    return 0;
}

constexpr int kGp1dLongNP = 768;        // rows of the global-scratch matrix of a long band (160..767 valid points)
constexpr int kGp1dLongGrid = 256;
constexpr size_t kGp1dLongBytes = (size_t)kGp1dLongGrid * gp_store_doubles(kGp1dLongNP) * 8;
constexpr int kGp1dThreads = 256;       // measured: 64 threads 10.1 s, 128: 7.2 s, 256: 6.8 s per 125 k objects
// ---- per-band 1-D GP (gp1d.hpp): one light curve per workgroup of kGp1dThreads threads, its bands g, r, i, z one after
// the other; the band's rows are an index list into the CSR slice (time order: file order when the rows
// are sorted, a rank sort otherwise); NP - 1 = most valid points of a band, ROWCAP = most rows of the object
// Bands of at most WNP - 1 valid points (the common case) are fitted by the four wavefronts of the workgroup
// side by side, each on a WNP-row matrix of its own with wave-level fences only; otherwise the bands run one
// after the other on the whole workgroup with the NP-row matrix.  Both layouts share one LDS buffer.
template <int WNP>
struct Gp1dWaveLds {
    Gp1dLds<WNP, 1> S;
    double K[gp_store_doubles(WNP)];
};
template <int NP, int NW>
struct Gp1dBlockLds {
    Gp1dLds<NP, NW> S;
    double K[gp_store_doubles(NP)];
};

// MW = waves per SIMD the register allocation leaves room for (= workgroups per CU at 256 threads)
template <int NP, int ROWCAP, int T, int WNP, int MW>
__global__ __launch_bounds__(T, MW) void gp1d_kernel(BatchView B, Bins bins, int bin, int nan_from, double* out, int ld,
                                                   int col0, int32_t* status, int st_ld, int st0,
                                                   unsigned long long* ticket, double* kslab) {
    using W = BlockDev<T>;
    constexpr bool kWavePath = (T == 256);
    constexpr size_t kBlockBytes = sizeof(Gp1dBlockLds<NP, W::NWAVES>), kWaveBytes = kWavePath ? 4 * sizeof(Gp1dWaveLds<WNP>) : 0;
    __shared__ __attribute__((aligned(16))) unsigned char raw[(kBlockBytes > kWaveBytes) ? kBlockBytes : kWaveBytes];
    auto& LB = *reinterpret_cast<Gp1dBlockLds<NP, W::NWAVES>*>(raw);
    __shared__ unsigned short rows[ROWCAP], rows2[ROWCAP];
    __shared__ double orow[GP1D_NCOL + 3];
    __shared__ int boff[5], nvalid[4];
    __shared__ int wcnt[T / 64][4];
    __shared__ long long next_ticket;
    const int count = bins.counts[kNumBins + bin];
    const int* list = bins.lists + (int64_t)(kNumBins + bin) * bins.stride;
    for (;;) {
        if (threadIdx.x == 0) next_ticket = (long long)atomicAdd(ticket, 1ull);
        __syncthreads();
        const int64_t pos = next_ticket;
        __syncthreads();
        if (pos >= count) break;
        const int64_t i = list[pos];
        const int64_t s = B.offsets[i];
        const int n = (int)(B.offsets[i + 1] - s);
        const double* t = B.t + s;
        const double* f = B.f + s;
        const double* e = B.e + s;
        const uint8_t* bb = B.b + s;
        int32_t* st = status ? status + i * (int64_t)st_ld + st0 : nullptr;
        // ---- rows of the bands g, r, i, z as one index list partitioned by band, each part in time order
        int cnt[4] = {0, 0, 0, 0}, nv[4] = {0, 0, 0, 0};
        for (int k = threadIdx.x; k < n; k += T) {
            const int band = bb[k];
            if (band < 1 || band > 4) continue;
            ++cnt[band - 1];
            if (!is_nan(f[k]) && !is_nan(e[k]) && e[k] > 0) ++nv[band - 1];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { cnt[j] = W::sum(cnt[j]); nv[j] = W::sum(nv[j]); }
        if (threadIdx.x == 0) {
            boff[0] = 0;
            for (int j = 0; j < 4; ++j) { boff[j + 1] = boff[j] + cnt[j]; nvalid[j] = nv[j]; }
        }
        __syncthreads();
        // stable partition by band: ballot ranks inside a wavefront, wave totals through LDS, running totals per chunk
        // (one pass over the band bytes instead of one scan of all earlier rows per row)
        {
            int run[4] = {0, 0, 0, 0};
            for (int base = 0; base < n; base += T) {
                const int k = base + (int)threadIdx.x;
                const int band = (k < n) ? (int)bb[k] : 0;
                int rank = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned long long mk = __ballot(band == j + 1);
                    if (band == j + 1) rank = WaveDev::prefix(mk);
                    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6][j] = popcll(mk);
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int before = 0, total = 0;
                    for (int wv = 0; wv < T / 64; ++wv) { const int c = wcnt[wv][j]; total += c; before += (wv < (int)(threadIdx.x >> 6)) ? c : 0; }
                    if (band == j + 1) rows[boff[j] + run[j] + before + rank] = (unsigned short)k;
                    run[j] += total;
                }
                __syncthreads();
            }
        }
        bool ordered = true;
        for (int k = threadIdx.x; k + 1 < n; k += T) ordered = ordered && (t[k] <= t[k + 1]);
        if (!W::all(ordered)) {
            // stable rank sort of every band part by (time, file index), as the oracle's ``band_sorted``
            for (int j = 0; j < 4; ++j) {
                const int b0 = boff[j], m = boff[j + 1] - b0;
                for (int k = threadIdx.x; k < m; k += T) {
                    const double tk = t[rows[b0 + k]];
                    int r = 0;
                    for (int q = 0; q < m; ++q) {
                        const double tq = t[rows[b0 + q]];
                        r += (tq < tk || (tq == tk && q < k)) ? 1 : 0;
                    }
                    rows2[b0 + r] = rows[b0 + k];
                }
            }
            __syncthreads();
            for (int k = threadIdx.x; k < boff[4]; k += T) rows[k] = rows2[k];
            __syncthreads();
        }
        bool fitted[4];
        int most = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { fitted[j] = (boff[j + 1] - boff[j]) >= 5; most = (nvalid[j] > most) ? nvalid[j] : most; }
        // phase 1: every band that fits a WNP-row matrix, one per wavefront, side by side
        if constexpr (kWavePath) {
            const int j = threadIdx.x >> 6;
            if (nvalid[j] + 1 <= WNP) {
                auto& LW = reinterpret_cast<Gp1dWaveLds<WNP>*>(raw)[j];
                const int b0 = boff[j], m = boff[j + 1] - b0;
                gp1d_band<WaveOfBlock, WNP>(
                    [&](int r, double& tt, double& ff, double& ee) { const int k = rows[b0 + r]; tt = t[k]; ff = f[k]; ee = e[k]; }, m, LW.S,
                    [&](const double* x, int nn, double& fv, double* gv) { gp1d_eval<WaveOfBlock, WNP, lds_double*>(x, nn, LW.S, (lds_double*)LW.K, fv, gv); },
                    orow + 4 * j, st ? st + j : nullptr);
            }
            __syncthreads();
        }
        // phase 2: the longer bands one after the other on the whole workgroup (same LDS buffer, NP-row matrix)
        if (!kWavePath || most + 1 > WNP) {
            for (int j = 0; j < 4; ++j) {
                if (kWavePath && nvalid[j] + 1 <= WNP) continue;
                const int b0 = boff[j], m = boff[j + 1] - b0;
                gp1d_band<W, NP>([&](int r, double& tt, double& ff, double& ee) { const int k = rows[b0 + r]; tt = t[k]; ff = f[k]; ee = e[k]; },
                                 m, LB.S,
                                 [&](const double* x, int nn, double& fv, double* gv) { gp1d_eval<W, NP, lds_double*>(x, nn, LB.S, (lds_double*)LB.K, fv, gv); },
                                 orow + 4 * j, st ? st + j : nullptr);
                __syncthreads();
            }
        }
        // phase 3 (tier of 160..767 rows only): a band with more valid points than the LDS matrix takes has its Gram matrix
        // in this workgroup's slab of global scratch; the working set of a kGp1dLongNP-row fit aliases the LDS buffer
        if constexpr (ROWCAP > NP) {
            if (kslab != nullptr && most + 1 > NP) {
                static_assert(sizeof(Gp1dLds<kGp1dLongNP, W::NWAVES>) <= sizeof(raw), "long-band working set must fit the LDS buffer");
                auto& LG = *reinterpret_cast<Gp1dLds<kGp1dLongNP, W::NWAVES>*>(raw);
                double* Kg = kslab + (size_t)blockIdx.x * (size_t)gp_store_doubles(kGp1dLongNP);
                for (int j = 0; j < 4; ++j) {
                    if (nvalid[j] + 1 <= NP) continue;
                    const int b0 = boff[j], m = boff[j + 1] - b0;
                    gp1d_band<W, kGp1dLongNP>([&](int r, double& tt, double& ff, double& ee) { const int k = rows[b0 + r]; tt = t[k]; ff = f[k]; ee = e[k]; },
                                              m, LG,
                                              [&](const double* x, int nn, double& fv, double* gv) { gp1d_eval<W, kGp1dLongNP, global_double*>(x, nn, LG, (global_double*)Kg, fv, gv); },
                                              orow + 4 * j, st ? st + j : nullptr);
                    __syncthreads();
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) gp1d_cross_band(orow, fitted);
        __syncthreads();
        store_row<W>(orow, out + i * (int64_t)ld + col0, GP1D_NCOL);
        __syncthreads();
    }
    nan_fill_bins<W>(bins, 1, nan_from, out, ld, col0, GP1D_NCOL, status, st_ld, st0, GP1D_NSTATUS);
}

template <int NP, int ROWCAP, int T, int WNP, int MW>
int launch_gp1d_tier(const BatchView& B, const Bins& bins, int bin, int nan_from, double* out, int ld, int col0,
                     int32_t* status, int st_ld, int st0, hipStream_t stream, int dev, unsigned long long* ticket,
                     double* kslab = nullptr) {
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gp1d_kernel<NP, ROWCAP, T, WNP, MW>, T, 0));
    if (per_cu < 1) per_cu = 1;
    int64_t grid = (int64_t)num_cus(dev) * per_cu;
    if (kslab && grid > kGp1dLongGrid) grid = kGp1dLongGrid;        // one slab of global scratch per workgroup
    if (grid > B.n_obj) grid = B.n_obj;
    if (grid < 1) return 0;
    hipLaunchKernelGGL((gp1d_kernel<NP, ROWCAP, T, WNP, MW>), dim3((unsigned)grid), dim3(T), 0, stream, B, bins, bin, nan_from, out, ld,
                       col0, status, st_ld, st0, ticket, kslab);
    HIP_TRY(hipGetLastError());
    return 0;
}

// The GP bins are reused (by rows of the whole object): bins 0..2 run with a Gram matrix as large as the
// object; bins 3..5 (160..767 rows) with the 160-row matrix -- a band with more than 159 valid points
// gets NaN and status -100.
int launch_gp1d(const BatchView& B, const Bins& bins, int64_t max_len, double* out, int ld, int col0, int32_t* status,
                int st_ld, int st0, hipStream_t stream, int dev, int* n_launch, unsigned long long* tickets, double* kslab) {
    const int caps[6] = {63, 111, 159, kGpSmallNP - 1, kGpMidNP - 1, kGpGlobalNP - 1};
    int last = 0;
    while (last < 5 && caps[last] < max_len) ++last;
    for (int ti = last; ti >= 0; --ti) {
        const int nan_from = (ti == last) ? ti + 1 : kNumBins;
        unsigned long long* tk = tickets + SET_GP1D * 8 + ti;
        int rc = 0;
        switch (ti) {
            case 0: rc = launch_gp1d_tier<64, 64, kGp1dThreads, 32, 3>(B, bins, ti, nan_from, out, ld, col0, status, st_ld, st0, stream, dev, tk); break;
            case 1: rc = launch_gp1d_tier<112, 112, kGp1dThreads, 32, 2>(B, bins, ti, nan_from, out, ld, col0, status, st_ld, st0, stream, dev, tk); break;
            case 2: rc = launch_gp1d_tier<160, 160, kGp1dThreads, 64, 1>(B, bins, ti, nan_from, out, ld, col0, status, st_ld, st0, stream, dev, tk); break;
            default: rc = launch_gp1d_tier<160, 768, kGp1dThreads, 64, 1>(B, bins, ti, nan_from, out, ld, col0, status, st_ld, st0, stream, dev, tk, kslab); break;
        }
        if (rc) return rc;
        ++*n_launch;
    }
    return 0;
}

const int kTiers[] = {128, 256, 512, 1024, 2048};
constexpr int kMaxPoints = 2048;

int g_num_cu[16] = {0};
std::mutex g_num_cu_mutex;

int num_cus(int dev) {
    if (dev < 0 || dev >= 16) return 256;
    std::lock_guard<std::mutex> lock(g_num_cu_mutex);
    if (!g_num_cu[dev]) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) g_num_cu[dev] = p.multiProcessorCount;
        else g_num_cu[dev] = 256;
    }
    return g_num_cu[dev];
}

template <int SET, int CAP>
int launch_tier(const BatchView& B, const Bins& bins, int bin, int nan_from, double* out, int ld, int col0,
                int32_t* status, int st_ld, int st0, hipStream_t stream, int dev, unsigned long long* ticket,
                int64_t max_grid = 1 << 30) {
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, set_kernel<SET, CAP>, 64, 0));
    if (per_cu < 1) per_cu = 1;
    int64_t grid = (int64_t)num_cus(dev) * per_cu;
    if (grid > max_grid) grid = max_grid;
    const int chunk = (SET == SET_BAZIN || SET == SET_POWERLAW) ? 1 : 8;
    if (grid * chunk > B.n_obj) grid = (B.n_obj + chunk - 1) / chunk;
    if (grid < 1) return 0;
    hipLaunchKernelGGL((set_kernel<SET, CAP>), dim3((unsigned)grid), dim3(64), 0, stream, B, bins, bin, nan_from,
                       out, ld, col0, status, st_ld, st0, ticket, chunk);
    HIP_TRY(hipGetLastError());
    return 0;
}

// largest LDS tier a set's working memory fits in (160 KiB per workgroup)
template <int SET>
constexpr int max_tier() {
    return (SET == SET_BAZIN || SET == SET_POWERLAW || SET == SET_RESEARCH) ? 3 : 4;
}

template <int SET>
int launch_set(const BatchView& B, const Bins& bins, int64_t max_len, double* out, int ld, int col0, int32_t* status,
               int st_ld, int st0, hipStream_t stream, int dev, int* n_launch, unsigned long long* tickets, char* long_slabs) {
    // tiers needed: every tier whose window (prev_cap, cap] can contain an object, i.e. up to the
    // first cap >= max_len; the last launched tier also NaN-fills the bins of longer objects.
    int last = 0;
    while (last < max_tier<SET>() && kTiers[last] < max_len) ++last;
    for (int ti = 0; ti <= last; ++ti) {
        const int nan_from = (ti == last) ? ti + 1 : kNumBins;
        unsigned long long* tk = tickets + SET * 8 + ti;
        int rc = 0;
        switch (ti) {
            case 0: rc = launch_tier<SET, 128>(B, bins, ti, nan_from, out, ld, col0, status, st_ld, st0, stream, dev, tk); break;
            case 1: rc = launch_tier<SET, 256>(B, bins, ti, nan_from, out, ld, col0, status, st_ld, st0, stream, dev, tk); break;
            case 2: rc = launch_tier<SET, 512>(B, bins, ti, nan_from, out, ld, col0, status, st_ld, st0, stream, dev, tk); break;
            case 3: rc = launch_tier<SET, 1024>(B, bins, ti, nan_from, out, ld, col0, status, st_ld, st0, stream, dev, tk); break;
            case 4:
                if constexpr (max_tier<SET>() >= 4)
                    rc = launch_tier<SET, 2048>(B, bins, ti, nan_from, out, ld, col0, status, st_ld, st0, stream, dev, tk);
                break;
        }
        if (rc) return rc;
        ++*n_launch;
    }
    // the long-object tier: bin 6 (more than 2048 rows); the research set's LDS tiers end at 1024 rows (bin 4 too) and at
    // 4096 days of r band (its overflow list)
    if (long_slabs && (max_len > kTiers[max_tier<SET>()] || SET == SET_RESEARCH)) {
        const int rc = launch_long<SET>(B, bins, 6, (SET == SET_RESEARCH) ? 4 : -1, (SET == SET_RESEARCH) ? kResearchLongList : -1, out, ld, col0,
                                        status, st_ld, st0, stream, tickets + SET * 8 + 7, long_slabs);
        if (rc) return rc;
        ++*n_launch;
    }
    return 0;
}

size_t bazin_ws_bytes(int64_t n_obj, int64_t n_points) {
    const size_t np = (size_t)(n_points > 0 ? n_points : 1), no = (size_t)(n_obj > 0 ? n_obj : 1);
    return ((3 * 8 * np + 255) & ~(size_t)255) + ((32 * no + 255) & ~(size_t)255) + ((4 * (size_t)kFitTiers * 6 * no + 255) & ~(size_t)255);
}

template <int CAP>
int launch_bazin_partition(const BatchView& B, const Bins& bins, int bin, int nan_from, const FitWs& F, double* out, int ld, int col0,
                           int32_t* status, int st_ld, int st0, hipStream_t stream, int dev, unsigned long long* ticket) {
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bazin_partition_kernel<CAP>, 64, 0));
    if (per_cu < 1) per_cu = 1;
    int64_t grid = (int64_t)num_cus(dev) * per_cu;
    if (grid * 8 > B.n_obj) grid = (B.n_obj + 7) / 8;
    if (grid < 1) return 0;
    hipLaunchKernelGGL((bazin_partition_kernel<CAP>), dim3((unsigned)grid), dim3(64), 0, stream, B, bins, bin, nan_from, F, out, ld, col0,
                       status, st_ld, st0, ticket, 8);
    HIP_TRY(hipGetLastError());
    return 0;
}

// tuning knob: LCFE_FIT_WAVES_<rows>=k caps the fit kernels of that band-length tier at k wavefronts per CU (their LDS
// regions otherwise take most of a CU and keep the LDS-resident GP tiers off it)
static int fit_waves_cap(int bcap, int per_cu) {
    char name[40];
    snprintf(name, sizeof name, "LCFE_FIT_WAVES_%d", bcap);
    const char* e = getenv(name);
    if (e && atoi(e) > 0 && atoi(e) < per_cu) return atoi(e);
    return per_cu;
}

template <int BCAP>
int launch_bazin_fits(const BatchView& B, const Bins& bins, const FitWs& F, int tier, double* out, int ld, int col0, int32_t* status,
                      int st_ld, int st0, hipStream_t stream, int dev, unsigned long long* ticket) {
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bazin_fit_kernel<BCAP>, 64, 0));
    if (per_cu < 1) per_cu = 1;
    per_cu = fit_waves_cap(BCAP, per_cu);
    int64_t grid = (int64_t)num_cus(dev) * per_cu;
    if (grid * 8 > 6 * B.n_obj) grid = (6 * B.n_obj + 7) / 8;
    if (grid < 1) return 0;
    hipLaunchKernelGGL((bazin_fit_kernel<BCAP>), dim3((unsigned)grid), dim3(64), 0, stream, B, bins, F, tier, out, ld, col0, status,
                       st_ld, st0, ticket);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Bazin: partition pass per object tier, the fit kernel per band-length tier (longest first), the object-level
// kernel for objects with a band beyond the largest fit tier, then the cross-band columns.
int launch_bazin(const BatchView& B, const Bins& bins, int64_t max_len, double* out, int ld, int col0, int32_t* status, int st_ld,
                 int st0, hipStream_t stream, int dev, int* n_launch, unsigned long long* tickets, void* ws, size_t ws_bytes,
                 int64_t n_points, char* long_slabs) {
    if (!ws || ws_bytes < bazin_ws_bytes(B.n_obj, n_points))
        return fail_msg("lcfe_extract_device: workspace too small for the Bazin fit lists");
    const size_t np = (size_t)(n_points > 0 ? n_points : 1), no = (size_t)B.n_obj;
    FitWs F;
    char* p = (char*)ws;
    F.pt = (double*)p; F.pf = F.pt + np; F.pe = F.pf + np;
    p += (3 * 8 * np + 255) & ~(size_t)255;
    F.pboff = (int*)p;
    p += (32 * no + 255) & ~(size_t)255;
    F.fits = (int*)p;
    F.fit_stride = 6 * (int64_t)no;
    int last = 0;
    while (last < 4 && kTiers[last] < max_len) ++last;
    unsigned long long* tk = tickets + SET_BAZIN * 8;
    for (int ti = 0; ti <= last; ++ti) {
        const int nan_from = (ti == last) ? ti + 1 : kNumBins;
        int rc = 0;
        switch (ti) {
            case 0: rc = launch_bazin_partition<128>(B, bins, ti, nan_from, F, out, ld, col0, status, st_ld, st0, stream, dev, tk + ti); break;
            case 1: rc = launch_bazin_partition<256>(B, bins, ti, nan_from, F, out, ld, col0, status, st_ld, st0, stream, dev, tk + ti); break;
            case 2: rc = launch_bazin_partition<512>(B, bins, ti, nan_from, F, out, ld, col0, status, st_ld, st0, stream, dev, tk + ti); break;
            case 3: rc = launch_bazin_partition<1024>(B, bins, ti, nan_from, F, out, ld, col0, status, st_ld, st0, stream, dev, tk + ti); break;
            case 4: rc = launch_bazin_partition<2048>(B, bins, ti, nan_from, F, out, ld, col0, status, st_ld, st0, stream, dev, tk + ti); break;
        }
        if (rc) return rc;
        ++*n_launch;
    }
    // objects with a band of more than 256 rows: the object-level kernel (all 52 columns).  A small grid, and AHEAD of the
    // fit tiers: the list is normally empty, but every workgroup of this kernel needs 136 KiB of LDS -- at the end of the
    // stream it sat in the dispatcher for 100-180 ms, until a GP tier released a whole CU (tools/step_timeline.py)
    int rc = launch_tier<SET_BAZIN, 1024>(B, bins, kBazinFallbackList, kNumBins, out, ld, col0, status, st_ld, st0, stream, dev, tk + 5, 32);
    if (rc) return rc;
    ++*n_launch;
    rc = launch_bazin_fits<256>(B, bins, F, 3, out, ld, col0, status, st_ld, st0, stream, dev, tickets + kFitTicketBase + 3);
    if (!rc) rc = launch_bazin_fits<128>(B, bins, F, 2, out, ld, col0, status, st_ld, st0, stream, dev, tickets + kFitTicketBase + 2);
    if (!rc) rc = launch_bazin_fits<64>(B, bins, F, 1, out, ld, col0, status, st_ld, st0, stream, dev, tickets + kFitTicketBase + 1);
    if (!rc) rc = launch_bazin_fits<32>(B, bins, F, 0, out, ld, col0, status, st_ld, st0, stream, dev, tickets + kFitTicketBase + 0);
    if (rc) return rc;
    *n_launch += 4;
    hipLaunchKernelGGL(bazin_cross_kernel, dim3((unsigned)((B.n_obj + 255) / 256)), dim3(256), 0, stream, B, out, ld, col0);
    HIP_TRY(hipGetLastError());
    ++*n_launch;
    // the long-object tier: more than 2048 rows (bin 6), or more than 1024 rows with a band beyond the fit tiers (all 52 columns)
    if (long_slabs && max_len > 1024) {
        rc = launch_long<SET_BAZIN>(B, bins, 6, kBazinLongList, -1, out, ld, col0, status, st_ld, st0, stream, tk + 7, long_slabs);
        if (rc) return rc;
        ++*n_launch;
    }
    return 0;
}

size_t powerlaw_ws_bytes(int64_t n_obj, int64_t n_points) {
    const size_t np = (size_t)(n_points > 0 ? n_points : 1), no = (size_t)(n_obj > 0 ? n_obj : 1);
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return al(2 * 8 * np) + al(2 * 3 * 8 * no) + al(32 * no) + al(12 * no) + al(4 * (size_t)kFitTiers * 21 * no) + al(4 * (size_t)kFitTiers * 6 * no);
}

template <int CAP>
int launch_powerlaw_partition(const BatchView& B, const Bins& bins, int bin, int nan_from, const PlWs& F, double* out, int ld, int col0,
                              int32_t* status, int st_ld, int st0, hipStream_t stream, int dev, unsigned long long* ticket) {
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, powerlaw_partition_kernel<CAP>, 64, 0));
    if (per_cu < 1) per_cu = 1;
    int64_t grid = (int64_t)num_cus(dev) * per_cu;
    if (grid * 8 > B.n_obj) grid = (B.n_obj + 7) / 8;
    if (grid < 1) return 0;
    hipLaunchKernelGGL((powerlaw_partition_kernel<CAP>), dim3((unsigned)grid), dim3(64), 0, stream, B, bins, bin, nan_from, F, out, ld,
                       col0, status, st_ld, st0, ticket, 8);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int N, int BCAP>
int launch_powerlaw_fits(const BatchView& B, const Bins& bins, const PlWs& F, int tier, double* out, int ld, int col0, int32_t* status,
                         int st_ld, int st0, hipStream_t stream, int dev, unsigned long long* ticket) {
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (powerlaw_fit_kernel<N, BCAP>), 64, 0));
    if (per_cu < 1) per_cu = 1;
    per_cu = fit_waves_cap(BCAP, per_cu);
    int64_t grid = (int64_t)num_cus(dev) * per_cu;
    const int64_t nfit = ((N == 2) ? 21 : 6) * B.n_obj;
    if (grid * 8 > nfit) grid = (nfit + 7) / 8;
    if (grid < 1) return 0;
    hipLaunchKernelGGL((powerlaw_fit_kernel<N, BCAP>), dim3((unsigned)grid), dim3(64), 0, stream, B, bins, F, tier, out, ld, col0, status,
                       st_ld, st0, ticket);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_powerlaw(const BatchView& B, const Bins& bins, int64_t max_len, double* out, int ld, int col0, int32_t* status, int st_ld,
                    int st0, hipStream_t stream, int dev, int* n_launch, unsigned long long* tickets, void* ws, size_t ws_bytes,
                    int64_t n_points, char* long_slabs) {
    if (!ws || ws_bytes < powerlaw_ws_bytes(B.n_obj, n_points))
        return fail_msg("lcfe_extract_device: workspace too small for the decline-fit lists");
    const size_t np = (size_t)(n_points > 0 ? n_points : 1), no = (size_t)B.n_obj;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    PlWs F;
    char* p = (char*)ws;
    F.tp = (double*)p; F.fp = F.tp + np; p += al(2 * 8 * np);
    F.peak = (double*)p; F.sstot = F.peak + 3 * no; p += al(2 * 3 * 8 * no);
    F.pboff = (int*)p; p += al(32 * no);
    F.kk = (int*)p; p += al(12 * no);
    F.fitsA = (int*)p; p += al(4 * (size_t)kFitTiers * 21 * no);
    F.fitsB = (int*)p;
    F.strideA = 21 * (int64_t)no;
    F.strideB = 6 * (int64_t)no;
    int last = 0;
    while (last < 4 && kTiers[last] < max_len) ++last;
    unsigned long long* tk = tickets + SET_POWERLAW * 8;
    for (int ti = 0; ti <= last; ++ti) {
        const int nan_from = (ti == last) ? ti + 1 : kNumBins;
        int rc = 0;
        switch (ti) {
            case 0: rc = launch_powerlaw_partition<128>(B, bins, ti, nan_from, F, out, ld, col0, status, st_ld, st0, stream, dev, tk + ti); break;
            case 1: rc = launch_powerlaw_partition<256>(B, bins, ti, nan_from, F, out, ld, col0, status, st_ld, st0, stream, dev, tk + ti); break;
            case 2: rc = launch_powerlaw_partition<512>(B, bins, ti, nan_from, F, out, ld, col0, status, st_ld, st0, stream, dev, tk + ti); break;
            case 3: rc = launch_powerlaw_partition<1024>(B, bins, ti, nan_from, F, out, ld, col0, status, st_ld, st0, stream, dev, tk + ti); break;
            case 4: rc = launch_powerlaw_partition<2048>(B, bins, ti, nan_from, F, out, ld, col0, status, st_ld, st0, stream, dev, tk + ti); break;
        }
        if (rc) return rc;
        ++*n_launch;
    }
    // the object-level kernel for the (normally empty) list of objects beyond the fit tiers, AHEAD of the fit tiers: see launch_bazin
    int rc = launch_tier<SET_POWERLAW, 1024>(B, bins, kPowerlawFallbackList, kNumBins, out, ld, col0, status, st_ld, st0, stream, dev, tk + 5, 32);
    if (rc) return rc;
    ++*n_launch;
#define PL_TIER(BC, T)                                                                                                        \
    if (!rc) rc = launch_powerlaw_fits<2, BC>(B, bins, F, T, out, ld, col0, status, st_ld, st0, stream, dev, tickets + kPlTicketA + T); \
    if (!rc) rc = launch_powerlaw_fits<3, BC>(B, bins, F, T, out, ld, col0, status, st_ld, st0, stream, dev, tickets + kPlTicketB + T);
    PL_TIER(256, 3) PL_TIER(128, 2) PL_TIER(64, 1) PL_TIER(32, 0)
#undef PL_TIER
    if (rc) return rc;
    *n_launch += 8;
    if (long_slabs && max_len > 1024) {
        rc = launch_long<SET_POWERLAW>(B, bins, 6, kPowerlawLongList, -1, out, ld, col0, status, st_ld, st0, stream, tk + 7, long_slabs);
        if (rc) return rc;
        ++*n_launch;
    }
    return 0;
}

template <int CAP>
int launch_stat_lean(const BatchView& B, const Bins& bins, int bin, double* out, int ld, int col0, hipStream_t stream,
                     int dev, unsigned long long* ticket, int64_t grid_cap = 0) {
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, stat_lean_kernel<CAP>, 64, 0));
    if (per_cu < 1) per_cu = 1;
    int64_t grid = (int64_t)num_cus(dev) * per_cu;
    if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;    // a list expected to be short: every wavefront's first ticket is an atomic on one counter
    if (grid * 8 > B.n_obj) grid = (B.n_obj + 7) / 8;
    if (grid < 1) return 0;
    hipLaunchKernelGGL((stat_lean_kernel<CAP>), dim3((unsigned)grid), dim3(64), 0, stream, B, bins, bin, out, ld, col0,
                       ticket, 8);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_stat_plan(const BatchView& B, const Bins& bins, hipStream_t stream) {
    const int64_t grid = (B.n_obj + kPlanThreads / 8 - 1) / (kPlanThreads / 8) + 3;
    hipLaunchKernelGGL(stat_plan_kernel, dim3((unsigned)grid), dim3(kPlanThreads), 0, stream, B, bins);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_stat_lanes_all(const BatchView& B, const Bins& bins, int retry, double* out, int ld, int col0, hipStream_t stream) {
    const int64_t grid = (B.n_obj + 3) / 4 + 6;
    hipLaunchKernelGGL(stat_lanes_all_kernel, dim3((unsigned)grid), dim3(64), 0, stream, B, bins, retry, out, ld, col0);
    HIP_TRY(hipGetLastError());
    return 0;
}

// LCFE_STAT_LANES=0 keeps the one-light-curve-per-wavefront kernels for the 128- and 256-row tiers (A/B measurements)
static bool stat_lanes_enabled() {
    const char* e = getenv("LCFE_STAT_LANES");
    return !(e && e[0] == '0');
}

// Statistics: lean kernels for the tiers up to 512 rows, the general kernel for the longer tiers,
// for the lean kernels' fallback list and for the NaN rows of over-long objects.
int launch_stat(const BatchView& B, const Bins& bins, int64_t max_len, double* out, int ld, int col0,
                hipStream_t stream, int dev, int* n_launch, unsigned long long* tickets, hipStream_t s1, hipStream_t s2, char* long_slabs) {
    int last = 0;
    while (last < 4 && kTiers[last] < max_len) ++last;
    unsigned long long* tk = tickets + SET_STAT * 8;
    // The tier kernels are independent (disjoint objects): with side streams they are enqueued side by side, so the
    // ramp-down of one tier is filled by the waves of the others; they are joined before the fallback launch.
    //   s2:     the tiers of more than 256 rows, from the start
    //   stream: the plan kernels (which lanes kernel takes a light curve of up to 256 rows), then the lanes kernels
    //   s1:     after the plan kernels, the 256-row tier's one-light-curve-per-wavefront kernel
    const bool lanes = stat_lanes_enabled();
    const bool fork = (s1 != stream) && (s2 != stream) && last >= 1;
    hipEvent_t ev_fork = nullptr, ev_plan = nullptr, ev_j1 = nullptr, ev_j2 = nullptr;
    struct Cleanup { hipEvent_t* e[4]; ~Cleanup() { for (auto p : e) if (*p) (void)hipEventDestroy(*p); } } cleanup{{&ev_fork, &ev_plan, &ev_j1, &ev_j2}};
    hipStream_t q_long = fork ? s2 : stream, q_mid = fork ? s1 : stream;
    if (fork) {
        HIP_TRY(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev_fork, stream));
        HIP_TRY(hipStreamWaitEvent(s2, ev_fork, 0));
    }
    for (int ti = 2; ti <= last; ++ti) {
        const int nan_from = (ti == last) ? ti + 1 : kNumBins;
        int rc = 0;
        switch (ti) {
            case 2: if (!lanes) rc = launch_stat_lean<512>(B, bins, ti, out, ld, col0, q_long, dev, tk + ti); break;   // (else: after its plan kernel)
            case 3: rc = launch_tier<SET_STAT, 1024>(B, bins, ti, nan_from, out, ld, col0, nullptr, 0, 0, q_long, dev, tk + ti); break;
            case 4: rc = launch_tier<SET_STAT, 2048>(B, bins, ti, nan_from, out, ld, col0, nullptr, 0, 0, q_long, dev, tk + ti); break;
        }
        if (rc) return rc;
        ++*n_launch;
    }
    if (lanes) {
        // one routing pass over the tiers of up to 512 rows (stat_plan_kernel): lanes lists + the retry list
        const int rc = launch_stat_plan(B, bins, stream);
        if (rc) return rc;
        ++*n_launch;
    }
    if (fork) {
        HIP_TRY(hipEventCreateWithFlags(&ev_plan, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev_plan, stream));
        HIP_TRY(hipStreamWaitEvent(s1, ev_plan, 0));
        if (lanes && last >= 2) HIP_TRY(hipStreamWaitEvent(s2, ev_plan, 0));
    }
    {
        int rc = 0;
        if (lanes) {
            // what the lanes variants do not take: one light curve per wavefront, sized for the longest tier present;
            // a light curve whose rows turn out not to ascend in time goes to the general kernel's list
            if (last >= 2) rc = launch_stat_lean<512>(B, bins, kStatRetryList, out, ld, col0, q_long, dev, tk + 2);
            else if (last == 1) rc = launch_stat_lean<256>(B, bins, kStatRetryList, out, ld, col0, q_mid, dev, tk + 1, 512);
            if (!rc) rc = launch_stat_lanes_all(B, bins, kStatFallbackList, out, ld, col0, stream);
            if (!rc && last < 1) rc = launch_stat_lean<128>(B, bins, kStatRetryList, out, ld, col0, stream, dev, tk + 6, 512);
            *n_launch += 2;
        } else {
            if (last >= 1) rc = launch_stat_lean<256>(B, bins, 1, out, ld, col0, q_mid, dev, tk + 1);
            if (!rc) rc = launch_stat_lean<128>(B, bins, 0, out, ld, col0, stream, dev, tk + 0);
            *n_launch += 2;
        }
        if (rc) return rc;
    }
    if (fork) {
        HIP_TRY(hipEventCreateWithFlags(&ev_j1, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_j2, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev_j1, s1));
        HIP_TRY(hipEventRecord(ev_j2, s2));
        HIP_TRY(hipStreamWaitEvent(stream, ev_j1, 0));
        HIP_TRY(hipStreamWaitEvent(stream, ev_j2, 0));
    }
    // fallback list of the lean tiers (+ the NaN rows when no general tier ran): normally empty or a
    // handful of objects, so a quarter-chip grid keeps the launch short
    const int nan_from = (last <= 2) ? last + 1 : kNumBins;
    const int64_t fb_grid = 256;
    int rc = 0;
    if (last == 0) rc = launch_tier<SET_STAT, 128>(B, bins, kStatFallbackList, nan_from, out, ld, col0, nullptr, 0, 0, stream, dev, tk + 5, fb_grid);
    else if (last == 1) rc = launch_tier<SET_STAT, 256>(B, bins, kStatFallbackList, nan_from, out, ld, col0, nullptr, 0, 0, stream, dev, tk + 5, fb_grid);
    else rc = launch_tier<SET_STAT, 512>(B, bins, kStatFallbackList, nan_from, out, ld, col0, nullptr, 0, 0, stream, dev, tk + 5, fb_grid);
    if (rc) return rc;
    ++*n_launch;
    if (long_slabs && max_len > kMaxPoints) {
        rc = launch_long<SET_STAT>(B, bins, 6, -1, -1, out, ld, col0, nullptr, 0, 0, stream, tk + 7, long_slabs);
        if (rc) return rc;
        ++*n_launch;
    }
    return 0;
}

#include "colnames.inc"

bool set_implemented(int set) { return set >= 0 && set < NUM_SETS; }

// Non-blocking side streams per device, created on first use and kept for the life of the process.
constexpr int kSideStreams = 5;       // [0] Bazin, [1] decline fits, [2] streaming sets + GP tiers, [3], [4] further GP tiers
hipStream_t g_side[16][kSideStreams];
bool g_side_ready[16] = {false};
std::mutex g_side_mutex;
int side_streams(int dev, hipStream_t* out) {
    if (dev < 0 || dev >= 16) return 1;
    std::lock_guard<std::mutex> lock(g_side_mutex);
    if (!g_side_ready[dev]) {
        for (int k = 0; k < kSideStreams; ++k)
            if (hipStreamCreateWithFlags(&g_side[dev][k], hipStreamNonBlocking) != hipSuccess) return 1;
        g_side_ready[dev] = true;
    }
    for (int k = 0; k < kSideStreams; ++k) out[k] = g_side[dev][k];
    return 0;
}

// staging buffers + events of the host-buffer entry point, per device
struct HostPathPool {
    static constexpr int NBUF = 9;     // offsets, t, flux, err, band, z, out, status, workspace
    void* buf[NBUF] = {};
    size_t cap[NBUF] = {};
    hipEvent_t ev[4] = {};
    std::mutex mutex;
};
HostPathPool g_pool[16];

}  // namespace

extern "C" {

#ifdef LCFE_TRF_PROF
// debug builds only: read and reset the TRF phase counters
int lcfe_debug_trf_prof(unsigned long long* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(lcfe::g_trf_prof), 64) != hipSuccess) return 1;
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(lcfe::g_trf_prof), z, 64) != hipSuccess) return 1;
    return 0;
}
#endif

#ifdef LCFE_DEBUG
// debug builds only: read and reset the counters of the bounds-checked lanes kernels
// (out4: indices outside the LDS buffer, damaged canary words, the last bad index, the length of its buffer)
int lcfe_debug_lanes_check(unsigned int* out4) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(lcfe::g_lanes_check), 16) != hipSuccess) return 1;
    unsigned int z[4] = {0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(lcfe::g_lanes_check), z, 16) != hipSuccess) return 1;
    return 0;
}
#endif

#ifdef LCFE_PHASE_PROF
// debug builds only: read and reset the phase cycle counters
int lcfe_debug_phase_prof(unsigned long long* out32) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(lcfe::g_phase_prof), 256) != hipSuccess) return 1;
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(lcfe::g_phase_prof), z, 256) != hipSuccess) return 1;
    return 0;
}
#endif

int lcfe_version(void) { return 1; }

int lcfe_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* lcfe_last_error(void) { return g_err.c_str(); }

int64_t lcfe_max_points(void) { return kLongCap; }

int64_t lcfe_gp2d_max_points(void) { return kGpLongNP - 1; }

int lcfe_implemented_mask(void) {
    int m = 0;
    for (int s = 0; s < NUM_SETS; ++s)
        if (set_implemented(s)) m |= 1 << s;
    return m;
}

int64_t lcfe_ncols(int mask) {
    int64_t n = 0;
    for (int s = 0; s < NUM_SETS; ++s)
        if (mask & (1 << s)) n += set_ncols(s);
    return n;
}

int64_t lcfe_nstatus(int mask) {
    int64_t n = 0;
    for (int s = 0; s < NUM_SETS; ++s)
        if (mask & (1 << s)) n += set_nstatus(s);
    return n;
}

const char* lcfe_colname(int mask, int64_t j) {
    if (j < 0) return nullptr;
    for (int s = 0; s < NUM_SETS; ++s) {
        if (!(mask & (1 << s))) continue;
        if (j < set_ncols(s)) return kColNames[s][j];
        j -= set_ncols(s);
    }
    return nullptr;
}

// workspace layout: [0, 1024) ticket counters (8 per set), [1024, 2048) bin counts, then the
// kNumLists index lists of n_obj int32 each (256-byte aligned total), then the GP scratch slabs
constexpr size_t kWsHeader = 2048;
static size_t list_bytes(int64_t n_obj) {
    return (((size_t)(n_obj > 0 ? n_obj : 0) * kNumLists * sizeof(int)) + 255) & ~(size_t)255;
}
size_t lcfe_workspace_bytes(int mask, int64_t n_obj, int64_t n_points) {
    size_t b = kWsHeader + list_bytes(n_obj);
    if (mask & (1 << SET_GP2D)) b += kGpScratchBytes;
    if (mask & (1 << SET_BAZIN)) b += bazin_ws_bytes(n_obj, n_points);
    if (mask & (1 << SET_POWERLAW)) b += powerlaw_ws_bytes(n_obj, n_points);
    if (mask & (1 << SET_GP1D)) b += kGp1dLongBytes;
    return b;
}

// slabs of the long-object tier of one set (0: the set needs none for light curves of up to max_len rows)
static size_t long_bytes_of(int set, int64_t max_len) {
    switch (set) {
        case SET_STAT: return (max_len > kMaxPoints) ? kLongGrid * long_slab_bytes<SET_STAT>() : 0;
        case SET_BAZIN: return (max_len > 1024) ? kLongGrid * long_slab_bytes<SET_BAZIN>() : 0;
        case SET_POWERLAW: return (max_len > 1024) ? kLongGrid * long_slab_bytes<SET_POWERLAW>() : 0;
        case SET_TDE: return (max_len > kMaxPoints) ? kLongGrid * long_slab_bytes<SET_TDE>() : 0;
        case SET_COLOR: return (max_len > kMaxPoints) ? kLongGrid * long_slab_bytes<SET_COLOR>() : 0;
        case SET_SHAPE: return (max_len > kMaxPoints) ? kLongGrid * long_slab_bytes<SET_SHAPE>() : 0;
        case SET_PHYSICS: return (max_len > kMaxPoints) ? kLongGrid * long_slab_bytes<SET_PHYSICS>() : 0;
        case SET_GP2D: return (max_len > kGpGlobalNP - 1) ? kGpLongBytes : 0;
        case SET_RESEARCH: return kLongGrid * long_slab_bytes<SET_RESEARCH>();     // (an r band of more than 4096 days can sit in a short light curve)
    }
    return 0;
}

size_t lcfe_workspace_bytes_for(int mask, int64_t n_obj, int64_t n_points, int64_t max_len) {
    size_t b = lcfe_workspace_bytes(mask, n_obj, n_points);
    for (int s = 0; s < NUM_SETS; ++s)
        if (mask & (1 << s)) b += long_bytes_of(s, max_len);
    return b;
}

int lcfe_extract_device(int mask, int device, void* stream_, int64_t n_obj, int64_t n_points,
                        int64_t max_len, const int64_t* d_offsets, const double* d_t, const double* d_flux,
                        const double* d_err, const uint8_t* d_band, const double* d_z, double* d_out,
                        int32_t* d_status, void* d_workspace, size_t workspace_bytes, lcfe_stats* prof) {
    g_err.clear();
    if (mask <= 0 || mask > LCFE_MASK_ALL) return fail_msg("lcfe_extract_device: empty or unknown feature-set mask");
    for (int s = 0; s < NUM_SETS; ++s)
        if ((mask & (1 << s)) && !set_implemented(s))
            return fail_msg("lcfe_extract_device: feature set " + std::to_string(s) + " is not built into this library");
    if (n_obj < 0 || n_points < 0 || max_len < 0) return fail_msg("lcfe_extract_device: negative size");
    if (n_obj > 0x7fffffff - kBinThreads) return fail_msg("lcfe_extract_device: more than 2^31 objects in one batch");
    // fit ids are packed into int32: object * 8 + band (Bazin), object * 32 + band * 9 + model (decline fits)
    if ((mask & (1 << SET_BAZIN)) && n_obj >= (1ll << 28))
        return fail_msg("lcfe_extract_device: the Bazin set takes fewer than 2^28 objects per batch (split the batch)");
    if ((mask & (1 << SET_POWERLAW)) && n_obj >= (1ll << 26))
        return fail_msg("lcfe_extract_device: the power-law set takes fewer than 2^26 objects per batch (split the batch)");
    if (n_obj == 0) return 0;
    if (!d_offsets || !d_out || (n_points > 0 && (!d_t || !d_flux || !d_err || !d_band)))
        return fail_msg("lcfe_extract_device: null array");
    DeviceGuard guard;
    if (guard.enter(device)) return fail_msg("lcfe_extract_device: cannot select device " + std::to_string(device));
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipStream_t stream = (hipStream_t)stream_;
    BatchView B{d_offsets, d_t, d_flux, d_err, d_band, d_z, n_obj};
    const int ld = (int)lcfe_ncols(mask);
    const int st_ld = (int)lcfe_nstatus(mask);
    if (prof) {
        memset(prof, 0, sizeof *prof);
        prof->bytes_in = 25 * n_points + 8 * (n_obj + 1) + (d_z ? 8 * n_obj : 0);
        prof->bytes_out = 8 * n_obj * (int64_t)ld;
    }
    const size_t lists_b = list_bytes(n_obj);
    if (!d_workspace || workspace_bytes < kWsHeader + lists_b)
        return fail_msg("lcfe_extract_device: workspace smaller than lcfe_workspace_bytes(mask, n_obj, n_points)");
    unsigned long long* tickets = (unsigned long long*)d_workspace;
    int* counts = (int*)((char*)d_workspace + 1024);
    int* lists = (int*)((char*)d_workspace + kWsHeader);
    if (workspace_bytes < lcfe_workspace_bytes(mask, n_obj, n_points))
        return fail_msg("lcfe_extract_device: workspace smaller than lcfe_workspace_bytes(mask, n_obj, n_points)");
    // after the lists: the GP scratch slabs (if the 2-D GP is in the mask), then the Bazin fit workspace
    char* region = (char*)d_workspace + kWsHeader + lists_b;
    const size_t gp_scratch_bytes = (mask & (1 << SET_GP2D)) ? kGpScratchBytes : 0;
    double* gp_scratch = gp_scratch_bytes ? (double*)region : nullptr;
    region += gp_scratch_bytes;
    void* bazin_ws = (mask & (1 << SET_BAZIN)) ? (void*)region : nullptr;
    const size_t bazin_bytes = bazin_ws ? bazin_ws_bytes(n_obj, n_points) : 0;
    region += bazin_bytes;
    void* pl_ws = (mask & (1 << SET_POWERLAW)) ? (void*)region : nullptr;
    const size_t pl_bytes = pl_ws ? powerlaw_ws_bytes(n_obj, n_points) : 0;
    region += pl_bytes;
    double* gp1d_slab = (mask & (1 << SET_GP1D)) ? (double*)region : nullptr;
    if (gp1d_slab) region += kGp1dLongBytes;
    // the slabs of the long-object tier, when the workspace was sized with lcfe_workspace_bytes_for(.., max_len)
    char* long_slab[NUM_SETS] = {};
    if (workspace_bytes >= lcfe_workspace_bytes_for(mask, n_obj, n_points, max_len)) {
        for (int s = 0; s < NUM_SETS; ++s) {
            const size_t lb = (mask & (1 << s)) ? long_bytes_of(s, max_len) : 0;
            if (lb) { long_slab[s] = region; region += lb; }
        }
    }
    const Bins bins{lists, counts, n_obj};
    // Launch plan.  The sets write disjoint columns and only read the bins, so after the shared prologue
    // (+ the statistics set, which stays alone so that its event time is a clean roofline sample) the
    // remaining sets are enqueued on side streams forked from the caller's stream and joined back
    // into it: the heavy-tailed end of one kernel is filled by the workgroups of another, and the
    // register-bound fit kernels (1-2 waves per SIMD) share the SIMDs with each other.
    // LCFE_SERIAL=1 keeps everything on the caller's stream (per-set times then do not overlap).
    static const bool serial = [] { const char* e = getenv("LCFE_SERIAL"); return e && e[0] == '1'; }();
    hipStream_t side[kSideStreams];
    const bool fork = !serial && side_streams(dev, side) == 0;
    // stream of each set: GP (the longest) stays on the caller's stream
    auto stream_of = [&](int s) -> hipStream_t {
        if (!fork) return stream;
        switch (s) {
            case SET_BAZIN: return side[0];
            case SET_POWERLAW: return side[1];
            case SET_TDE: case SET_COLOR: case SET_SHAPE: case SET_PHYSICS: case SET_GP1D: case SET_RESEARCH: return side[2];
            default: return stream;
        }
    };
    // timing events (prof only): created once per host thread and device, reused by later calls -- a call with
    // `prof` drains the stream before it returns, so the events of the previous call are always complete
    struct Events {
        hipEvent_t ev0[NUM_SETS] = {}, ev1[NUM_SETS] = {};
        bool ready = false;
    };
    static thread_local Events pools[16];
    Events& E = pools[(dev >= 0 && dev < 16) ? dev : 0];
    hipEvent_t (&ev0)[NUM_SETS] = E.ev0;
    hipEvent_t (&ev1)[NUM_SETS] = E.ev1;
    // fork / join markers are short-lived: an event that was recorded is released by the runtime once the
    // recorded work has completed, so destroying it right after the wait was enqueued is safe
    struct Marker {
        hipEvent_t e = nullptr;
        ~Marker() { if (e) (void)hipEventDestroy(e); }
    } fork_marker;
    hipEvent_t& forked = fork_marker.e;
    if (prof && !E.ready) {
        for (int k = 0; k < NUM_SETS; ++k) { HIP_TRY(hipEventCreate(&ev0[k])); HIP_TRY(hipEventCreate(&ev1[k])); }
        E.ready = true;
    }
    bool side_used[kSideStreams] = {false, false, false, false, false};
    int col0 = 0, st0 = 0, ne = 0;
    for (int s = 0; s < NUM_SETS; ++s) {
        if (!(mask & (1 << s))) continue;
        if (ne == 0) {
            // shared prologue (timed with the first set): zero tickets and counts, bin the objects
            if (prof) HIP_TRY(hipEventRecord(ev0[s], stream));
            HIP_TRY(hipMemsetAsync(d_workspace, 0, kWsHeader, stream));
            hipLaunchKernelGGL(bin_kernel, dim3((unsigned)((n_obj + kBinThreads - 1) / kBinThreads)), dim3(kBinThreads), 0,
                               stream, d_offsets, n_obj, lists, counts);
            HIP_TRY(hipGetLastError());
        }
        hipStream_t q = stream_of(s);
        if (fork && s != SET_STAT && !forked) {
            HIP_TRY(hipEventCreateWithFlags(&forked, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(forked, stream));
        }
        if (q != stream) {
            const int k = (q == side[0]) ? 0 : (q == side[1]) ? 1 : 2;
            if (!side_used[k]) { HIP_TRY(hipStreamWaitEvent(q, forked, 0)); side_used[k] = true; }
        }
        if (prof && ne != 0) HIP_TRY(hipEventRecord(ev0[s], q));
        int nl = 0, rc = 0;
        switch (s) {
            case SET_STAT: rc = launch_stat(B, bins, max_len, d_out, ld, col0, q, dev, &nl, tickets, fork ? side[0] : q, fork ? side[1] : q, long_slab[s]); break;
            case SET_BAZIN: rc = launch_bazin(B, bins, max_len, d_out, ld, col0, d_status, st_ld, st0, q, dev, &nl, tickets, bazin_ws, bazin_bytes, n_points, long_slab[s]); break;
            case SET_POWERLAW: rc = launch_powerlaw(B, bins, max_len, d_out, ld, col0, d_status, st_ld, st0, q, dev, &nl, tickets, pl_ws, pl_bytes, n_points, long_slab[s]); break;
            case SET_TDE: rc = launch_set<SET_TDE>(B, bins, max_len, d_out, ld, col0, d_status, st_ld, st0, q, dev, &nl, tickets, long_slab[s]); break;
            case SET_COLOR: rc = launch_set<SET_COLOR>(B, bins, max_len, d_out, ld, col0, d_status, st_ld, st0, q, dev, &nl, tickets, long_slab[s]); break;
            case SET_SHAPE: rc = launch_set<SET_SHAPE>(B, bins, max_len, d_out, ld, col0, d_status, st_ld, st0, q, dev, &nl, tickets, long_slab[s]); break;
            case SET_PHYSICS: rc = launch_set<SET_PHYSICS>(B, bins, max_len, d_out, ld, col0, d_status, st_ld, st0, q, dev, &nl, tickets, long_slab[s]); break;
            case SET_RESEARCH: rc = launch_set<SET_RESEARCH>(B, bins, max_len, d_out, ld, col0, d_status, st_ld, st0, q, dev, &nl, tickets, long_slab[s]); break;
            case SET_GP1D: rc = launch_gp1d(B, bins, max_len, d_out, ld, col0, d_status, st_ld, st0, q, dev, &nl, tickets, gp1d_slab); break;
            case SET_GP2D:
            {
                // the GP tiers, longest first, round-robin over the caller's stream and side streams 2.. (LCFE_GP_STREAMS, default
                // 2: more streams start more tiers at once)
                static const int want = [] { const char* e = getenv("LCFE_GP_STREAMS"); const int k = e ? atoi(e) : 2; return (k < 1) ? 1 : ((k > 4) ? 4 : k); }();
                hipStream_t gs[4] = {q, q, q, q};
                int ngs = 1;
                if (fork) {
                    for (int k = 1; k < want; ++k) {
                        gs[k] = side[1 + k];
                        if (!side_used[1 + k]) { HIP_TRY(hipStreamWaitEvent(side[1 + k], forked, 0)); side_used[1 + k] = true; }
                    }
                    ngs = want;
                }
                rc = launch_gp(B, bins, max_len, d_out, ld, col0, d_status, st_ld, st0, gs, ngs, dev, gp_scratch,
                               gp_scratch_bytes, &nl, tickets, long_slab[s]);
                // the set's stop event (prof) is recorded on q: make q wait for the tiers on the other streams
                for (int k = 1; k < ngs && !rc; ++k) {
                    hipEvent_t half;
                    HIP_TRY(hipEventCreateWithFlags(&half, hipEventDisableTiming));
                    HIP_TRY(hipEventRecord(half, gs[k]));
                    HIP_TRY(hipStreamWaitEvent(q, half, 0));
                    (void)hipEventDestroy(half);
                }
                break;
            }
        }
        if (rc) return rc;
        if (prof) { HIP_TRY(hipEventRecord(ev1[s], q)); prof->launches[s] = nl; }
        ++ne;
        col0 += set_ncols(s);
        st0 += set_nstatus(s);
    }
    // join the side streams back into the caller's stream
    for (int k = 0; k < kSideStreams; ++k) {
        if (!side_used[k]) continue;
        hipEvent_t done;
        HIP_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(done, side[k]));
        HIP_TRY(hipStreamWaitEvent(stream, done, 0));
        (void)hipEventDestroy(done);       // released once the recorded work has completed
    }
    if (prof) {
        HIP_TRY(hipStreamSynchronize(stream));
        for (int s = 0; s < NUM_SETS; ++s) {
            if (!(mask & (1 << s))) continue;
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, ev0[s], ev1[s]));
            prof->kernel_ms[s] = ms;
        }
    }
    return 0;
}

int lcfe_extract(int mask, int device, int64_t n_obj, const int64_t* offsets, const double* t,
                 const double* flux, const double* err, const uint8_t* band, const double* z, double* out,
                 int32_t* status, lcfe_stats* prof) {
    g_err.clear();
    if (n_obj < 0) return fail_msg("lcfe_extract: negative n_obj");
    if (n_obj == 0) return 0;
    if (!offsets || !out) return fail_msg("lcfe_extract: null array");
    if (lcfe_device_count() < 1) return fail_msg("lcfe_extract: no HIP device visible (liblcfe has no CPU fallback)");
    // host-side validation: a malformed CSR must never reach a kernel
    if (offsets[0] != 0) return fail_msg("lcfe_extract: offsets[0] != 0");
    int64_t max_len = 0;
    for (int64_t i = 0; i < n_obj; ++i) {
        const int64_t n = offsets[i + 1] - offsets[i];
        if (n < 0) return fail_msg("lcfe_extract: offsets not non-decreasing");
        if (n > max_len) max_len = n;
    }
    const int64_t np = offsets[n_obj];
    if (np > 0 && (!t || !flux || !err || !band)) return fail_msg("lcfe_extract: null sample array");
    const int64_t ld = lcfe_ncols(mask), st_ld = lcfe_nstatus(mask);
    if (ld <= 0) return fail_msg("lcfe_extract: empty feature-set mask");
    DeviceGuard guard;
    if (guard.enter(device)) return fail_msg("lcfe_extract: cannot select device " + std::to_string(device));
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return fail_msg("lcfe_extract: device index out of range");
    // Per-device pool of staging buffers and events, grown on demand and kept for the life of the process: a
    // statistics-only call used to spend 27 of its 38 ms in nine hipMalloc/hipFree pairs.  One host-buffer call
    // at a time per device holds the pool (concurrent callers on one device are serialised here).
    HostPathPool& P = g_pool[dev];
    std::lock_guard<std::mutex> pool_lock(P.mutex);
    const size_t npa = (size_t)(np > 0 ? np : 1);
    const size_t wsb = lcfe_workspace_bytes_for(mask, n_obj, np, max_len);
    const size_t need[HostPathPool::NBUF] = {sizeof(int64_t) * (size_t)(n_obj + 1), 8 * npa, 8 * npa, 8 * npa, npa,
                                             z ? 8 * (size_t)n_obj : 0, 8 * (size_t)n_obj * (size_t)ld,
                                             st_ld > 0 ? 4 * (size_t)n_obj * (size_t)st_ld : 0, wsb};
    for (int k = 0; k < HostPathPool::NBUF; ++k) {
        if (need[k] <= P.cap[k]) continue;
        if (P.buf[k]) { (void)hipFree(P.buf[k]); P.buf[k] = nullptr; P.cap[k] = 0; }
        const size_t want = need[k] + need[k] / 4;               // head room: batches of similar size reuse the block
        hipError_t e = hipMalloc(&P.buf[k], want);
        if (e != hipSuccess) { e = hipMalloc(&P.buf[k], need[k]); if (e == hipSuccess) P.cap[k] = need[k]; }
        else P.cap[k] = want;
        if (e != hipSuccess) return fail("hipMalloc(host-path pool)", e, __FILE__, __LINE__);
    }
    for (int k = 0; k < 4; ++k)
        if (!P.ev[k]) HIP_TRY(hipEventCreate(&P.ev[k]));
    int64_t* d_off = (int64_t*)P.buf[0];
    double *d_t = (double*)P.buf[1], *d_f = (double*)P.buf[2], *d_e = (double*)P.buf[3];
    uint8_t* d_b = (uint8_t*)P.buf[4];
    double* d_z = z ? (double*)P.buf[5] : nullptr;
    double* d_out = (double*)P.buf[6];
    int32_t* d_st = st_ld > 0 ? (int32_t*)P.buf[7] : nullptr;
    void* d_ws = P.buf[8];
    HIP_TRY(hipEventRecord(P.ev[0], 0));
    HIP_TRY(hipMemcpyAsync(d_off, offsets, sizeof(int64_t) * (n_obj + 1), hipMemcpyHostToDevice, 0));
    if (np > 0) {
        HIP_TRY(hipMemcpyAsync(d_t, t, 8 * np, hipMemcpyHostToDevice, 0));
        HIP_TRY(hipMemcpyAsync(d_f, flux, 8 * np, hipMemcpyHostToDevice, 0));
        HIP_TRY(hipMemcpyAsync(d_e, err, 8 * np, hipMemcpyHostToDevice, 0));
        HIP_TRY(hipMemcpyAsync(d_b, band, np, hipMemcpyHostToDevice, 0));
    }
    if (z) HIP_TRY(hipMemcpyAsync(d_z, z, 8 * n_obj, hipMemcpyHostToDevice, 0));
    if (d_st) HIP_TRY(hipMemsetAsync(d_st, 0, 4 * (size_t)n_obj * st_ld, 0));
    HIP_TRY(hipEventRecord(P.ev[1], 0));
    lcfe_stats local;
    const int rc = lcfe_extract_device(mask, -1, nullptr, n_obj, np, max_len, d_off, d_t, d_f, d_e, d_b, d_z, d_out, d_st,
                                       d_ws, wsb, prof ? &local : nullptr);
    if (rc) { (void)hipStreamSynchronize(0); return rc; }
    HIP_TRY(hipEventRecord(P.ev[2], 0));
    HIP_TRY(hipMemcpyAsync(out, d_out, 8 * (size_t)n_obj * ld, hipMemcpyDeviceToHost, 0));
    if (status && d_st) HIP_TRY(hipMemcpyAsync(status, d_st, 4 * (size_t)n_obj * st_ld, hipMemcpyDeviceToHost, 0));
    HIP_TRY(hipEventRecord(P.ev[3], 0));
    HIP_TRY(hipStreamSynchronize(0));
    if (prof) {
        *prof = local;
        float ms = 0;
        (void)hipEventElapsedTime(&ms, P.ev[0], P.ev[1]); prof->h2d_ms = ms;
        (void)hipEventElapsedTime(&ms, P.ev[2], P.ev[3]); prof->d2h_ms = ms;
    }
    return 0;
}

/* release the staging buffers lcfe_extract keeps per device (optional; they are reused by later calls) */
void lcfe_release_buffers(void) {
    for (int d = 0; d < 16; ++d) {
        HostPathPool& P = g_pool[d];
        std::lock_guard<std::mutex> lock(P.mutex);
        bool any = false;
        for (int k = 0; k < HostPathPool::NBUF; ++k) any = any || P.buf[k];
        if (!any) continue;
        DeviceGuard guard;
        if (guard.enter(d)) continue;
        for (int k = 0; k < HostPathPool::NBUF; ++k) { if (P.buf[k]) (void)hipFree(P.buf[k]); P.buf[k] = nullptr; P.cap[k] = 0; }
    }
}

}  // extern "C"
