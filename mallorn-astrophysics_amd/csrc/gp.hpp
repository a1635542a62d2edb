// gp.hpp -- 2-D (time x wavelength) Matern-3/2 Gaussian-process features
// (reference: src/features/multiband_gp.py; george semantics restated as in oracle/gp2d.py --
// PARITY UNPINNED vs the real george, which is not installed anywhere in the build).
//
// One light curve per workgroup.  The N x N Gram matrix (N = valid points of all bands) is kept as
// 16 x 16 lower-triangle tiles (fp64) in LDS or, for longer light curves, in a per-workgroup slab of
// global scratch, and is turned in place into -K^-1 by a blocked symmetric sweep whose row weights
// and rank-8/16 tile updates run on the fp64 MFMA (v_mfma_f64_16x16x4_f64); the sweep also yields
// log|K| and alpha = K^-1 r (augmented row), so every L-BFGS-B evaluation -- log-likelihood and
// analytic gradient -- costs one sweep plus one pass over the tiles.
#pragma once
#include "color.hpp"     // compute_color
#include "fits.hpp"      // wave_median
#include "lbfgsb.hpp"
#include "stage.hpp"

namespace lcfe {

constexpr int GP_NCOL = 27;
constexpr double GP_TINY = 1.25e-12;                        // george.gp.TINY (white noise floor)
constexpr double GP_LOG_2PI = 1.8378770664093453;

// Storage of the symmetric matrix: lower-triangular 16 x 16 TILES, each tile row-major and
// contiguous (2 KiB), tiles ordered (ti, tj <= ti) row by row.  Diagonal tiles keep their (unused)
// upper half.  This is the layout v_mfma_f64_16x16x4_f64 reads and writes with conflict-free
// 512-byte LDS accesses, and a tile is one coalesced 2 KiB piece of global scratch.
LCFE_HD int tile_base(int ti, int tj) { return (ti * (ti + 1) / 2 + tj) * 256; }            // tj <= ti
LCFE_HD int tri_index(int i, int j) { return tile_base(i >> 4, j >> 4) + ((i & 15) << 4) + (j & 15); }   // j <= i
LCFE_HD constexpr int gp_store_doubles(int np) { return ((np + 15) / 16) * ((np + 15) / 16 + 1) / 2 * 256; }

#if defined(__HIPCC__)
typedef __attribute__((address_space(3))) double lds_double;     // LDS-qualified element type: keeps ds_* addressing across calls
// global-scratch matrix: without the qualifier the (not inlined) evaluation sees a generic pointer and every tile access is a
// FLAT operation, which counts on the LDS counter as well -- each wait for an LDS read then also waits for the tile loads in flight
typedef __attribute__((address_space(1))) double global_double;
#endif

template <class P> struct gp_is_lds_ptr { static constexpr bool value = false; };
#if defined(__HIPCC__)
template <> struct gp_is_lds_ptr<lds_double*> { static constexpr bool value = true; };
#endif

// The sweep pivots one diagonal TILE (16 indices) per step.
constexpr int GP_B = 16;
// pivot-block width kept for code that sizes scratch by it
template <int NP> struct gp_block { static constexpr int B = GP_B; };
// row length of the pivot-column panel: NP rounded so that two consecutive k-slices of an MFMA operand read
// (rows 4kc+lr, lr = 0, 1) fall into different halves of the 64 LDS banks
LCFE_HD constexpr int gp_panel_ld(int np) { return ((np % 32) == 16) ? np : np + 16; }

// Working memory of one object; NP = capacity in rows (valid points + the augmented residual row).  The
// tile-packed matrix itself lives in LDS for the small tiers and in a per-workgroup slab of global scratch
// otherwise.
// second pivot panel + pivot inverse of the fused (two pivot tiles per pass) sweep of the global-scratch tiers
template <int NP, bool FUSE> struct GpFusePanels {};
template <int NP> struct GpFusePanels<NP, true> {
    double V2[GP_B][gp_panel_ld(NP)];
    double P2[GP_B][GP_B];
};

LCFE_FN double gp_wavelength(int band) {
    // multiband_gp.py:26-29
    const double w[6] = {3670.0, 4825.0, 6222.0, 7545.0, 8691.0, 9710.0};
    return w[band];
}

// wavelength per valid point and the scratch vector of the flux-scale median.  Plain arrays by default; the fused
// tiers keep the band code (one byte) instead of the wavelength and borrow the first pivot panel as scratch -- with
// their second panel that is what lets a 511-row light curve fit the 160 KiB of LDS.
template <int NP, int LDV, bool SLIM> struct GpSmallArrays {
    double lam[NP];
    double r[NP];
    LCFE_FN double lam_at(int i) const { return lam[i]; }
    LCFE_FN void set_band(int i, int band) { lam[i] = gp_wavelength(band); }
    LCFE_FN double* rbuf(double (*)[LDV]) { return r; }
};
template <int NP, int LDV> struct GpSmallArrays<NP, LDV, true> {
    unsigned char bnd[NP];
    LCFE_FN double lam_at(int i) const { const int b = bnd[i]; return gp_wavelength((b < 6) ? b : 0); }   // rows beyond the points hold stale bytes
    LCFE_FN void set_band(int i, int band) { bnd[i] = (unsigned char)band; }
    LCFE_FN double* rbuf(double (*V)[LDV]) { return &V[0][0]; }
};

// staging tile of the pivot look-ahead (workgroups of four or more wavefronts; a single wavefront has nobody to overlap with)
template <bool ON> struct GpStageTile { double stage[1]; };
template <> struct GpStageTile<true> { double stage[GP_B * GP_B]; };

template <int NP, int NW = 4, bool FUSE = false, bool LOOKAHEAD = true, bool IN_LDS = true>
struct GpLds {
    static constexpr bool kFuse = FUSE;
    static constexpr bool kLookAhead = LOOKAHEAD && NW >= 4;
    static constexpr bool kInLds = IN_LDS;    // false: the long-object tier keeps this working set in global scratch
    double t[NP], y[NP], e2[NP];              // valid points: time (from first valid), flux/scale, (err/scale)^2
    double alpha[NP];                         // K^-1 r
    GpSmallArrays<NP, gp_panel_ld(NP), FUSE> sm;
    double V[GP_B][gp_panel_ld(NP)];          // pivot-tile columns A(:, P) (rows p >= bs of a partial block are zero)
    double P[GP_B][GP_B];                     // inverse of the (identity-padded) pivot block
    GpFusePanels<NP, FUSE> fz;
    LbState<4, 10> lb;                        // L-BFGS-B state machine: advanced by thread 0 between two barriers
    int lb_why;
    double slot[2];
    double out[GP_NCOL + 1];
    int pivot_bad;                            // wave 0 reports a non-positive pivot
    int row_ticket;                           // tile rows of an update phase are handed out longest first
    GpStageTile<(LOOKAHEAD && NW >= 4)> la;                // wavefront 0: layout changes of the pivot look-ahead (one tile)
#ifdef LCFE_GP_PROF
    unsigned long long prof[12];
#endif
};

#if defined(LCFE_GP_PROF) && defined(__HIPCC__)
#define GP_T0() unsigned long long t0__ = __builtin_readcyclecounter()
#define GP_T(slot_) do { if (W::lane() == 0) { unsigned long long t1__ = __builtin_readcyclecounter(); S.prof[slot_] += t1__ - t0__; t0__ = t1__; } else { t0__ = 0; } } while (0)
#else
#define GP_T0() do {} while (0)
#define GP_T(slot_) do {} while (0)
#endif


#if defined(__HIPCC__)
typedef double gp_v4f64 __attribute__((ext_vector_type(4)));

// 1 / d: hardware reciprocal + two Newton steps (within an ulp of the division)
__device__ __forceinline__ double gp_recip(double d) {
    double inv = __builtin_amdgcn_rcp(d);
    inv = fma(fma(-d, inv, 1.0), inv, inv);
    return fma(fma(-d, inv, 1.0), inv, inv);
}

// One Gauss-Jordan pivot of the 16 x 16 block held by ONE wavefront, four elements per lane:
// lane l = 4 a + c keeps P(a, 4c .. 4c+3).  The pivot column entry (a, Q) comes from the lane's own quad
// (DPP quad broadcast), the pivot row entries (Q, 4c+r) from lane 4Q + c (LDS crossbar), the pivot from
// a v_readlane: ~40 instructions per pivot, against ~130 for a column-per-lane layout.
template <int Q>
__device__ __forceinline__ void gp_inv16_pivot(double (&p)[4], int a, int c, bool& bad, double& prod) {
    constexpr int QC = Q >> 2, QR = Q & 3;
    const double d = WaveDev::rdlane(p[QR], 4 * Q + QC);
    if (!(d > 0.0)) bad = true;
    prod *= d;
    const double inv = gp_recip(d);
    const double paq = WaveDev::dpp<QC * 0x55>(p[QR]);            // quad_perm(QC, QC, QC, QC): (a, Q)
    double t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = __shfl(p[r], 4 * Q + c, 64) * inv;    // (Q, 4c+r) / d
    const bool row_q = (a == Q), col_q = (c == QC);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double v = fma(-paq, t[r], p[r]);
        v = row_q ? t[r] : v;
        if (r == QR) v = col_q ? (row_q ? -inv : paq * inv) : v;
        p[r] = v;
    }
}
template <int Q>
__device__ __forceinline__ void gp_inv16_pivots(double (&p)[4], int a, int c, bool& bad, double& prod, double& ld) {
    if constexpr (Q < 16) {
        gp_inv16_pivot<Q>(p, a, c, bad, prod);
        if constexpr ((Q & 7) == 7) {
            // keep the running product in range: split off its exponent (the logarithm is taken once per sweep)
            const int e = __builtin_amdgcn_frexp_exp(prod);
            prod = __builtin_amdgcn_frexp_mant(prod);
            ld += (double)e;
        }
        gp_inv16_pivots<Q + 1>(p, a, c, bad, prod, ld);
    }
}
#endif

// rank of tile row i in the snake that deals the rows (longest first) to the NW wavefronts
LCFE_FN int gp_row_owner(int i, int nt, int nw) {
    const int k = nt - 1 - i, blk = k / nw, pos = k - blk * nw;
    return (blk & 1) ? nw - 1 - pos : pos;
}

// In-place inverse of the tile-packed symmetric positive-definite matrix A by a BLOCKED SYMMETRIC SWEEP: for
// each pivot block P (one diagonal tile, 16 consecutive indices; the last block may be partial)
//     A_PP <- -A_PP^-1 ,  A_RP <- A_RP A_PP^-1 ,  A_RR <- A_RR - A_RP A_PP^-1 A_PR     (R = all other indices)
// After all blocks A = -K^-1.  The pivots met while inverting the diagonal blocks are exactly the
// Cholesky pivots L_jj^2, so log|K| = sum log(pivot) and "pivot <= 0" is LAPACK dpotrf's failure
// (george: log-likelihood = -inf).
// The matrix carries one extra, never pivoted row n (the residual r = y - mu): the sweep turns it
// into alpha = K^-1 r and its diagonal into -r'K^-1 r (regression use of the sweep operator), so
// no solve or matrix-vector product is needed afterwards.
//
// GPU schedule.  Every wavefront OWNS whole tile rows (dealt longest-first in a snake, so the loads are even)
// and is the only writer of their tiles.  Per step: (1) the owners copy the pivot tile column into the panel
// V (LDS), (2) wavefront 0 inverts the pivot block in registers, (3) every wavefront forms, for each of its
// rows i, W_i = V_i D^-1 with four fp64 MFMAs -- computed as (D^-1 V_i')' so that the result lands in the
// registers in A-operand layout and is used at once, never stored as a panel --, writes it into the pivot
// column / row tiles, and applies C_ij -= W_i V_j' to its tiles (four v_mfma_f64_16x16x4_f64 per tile, four
// tiles in flight).  Three workgroup barriers per step, no per-element index arithmetic.
template <class W, int NP, class KP, class LDS>
LCFE_FN bool gp_sweep_inverse(KP A, int n, LDS& S, double& logdet) {
    const int nrow = n + 1;
    const int nt = (nrow + 15) >> 4;
#if defined(__HIPCC__)
    if constexpr (W::WAVE == 64) {
        constexpr int NW = W::NWAVES;
        constexpr int LDV = gp_panel_ld(NP);
        const int l = W::wlane(), lr = l >> 4, lc = l & 15, w = W::wave_id();
        double ld = 0.0, prod = 1.0;          // wavefront 0: exponent sum and mantissa product of the pivots

        // (1) pivot tile column kt -> Vp[p][i] = A(i, 16 kt + p); rows p >= bs of the panel are zero.  Tiles below the
        //     pivot are copied (transposed) by the owner of their row, the tiles of the pivot row itself by all
        //     wavefronts in turn; two tiles per trip so that their loads overlap.
        auto gather = [&](double (*Vp)[LDV], int kt, int bs) {
            const int k0 = kt << 4;
            for (int blk = 0;; blk += 2) {
                int ii[2];
                double x[2][4];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int pos = ((blk + u) & 1) ? NW - 1 - w : w;
                    ii[u] = nt - 1 - ((blk + u) * NW + pos);
                    KP T = A + tile_base((ii[u] > kt) ? ii[u] : kt, kt);
#pragma unroll
                    for (int v = 0; v < 4; ++v) x[u][v] = T[((lr + 4 * v) << 4) + lc];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (ii[u] > kt) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) Vp[lc][(ii[u] << 4) + lr + 4 * v] = (lc < bs) ? x[u][v] : 0.0;
                    }
                if (ii[1] <= kt) break;
            }
            for (int j0 = w; j0 < kt; j0 += 2 * NW) {
                double x[2][4];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int j = (j0 + u * NW < kt) ? j0 + u * NW : j0;
                    KP T = A + tile_base(kt, j);
#pragma unroll
                    for (int v = 0; v < 4; ++v) x[u][v] = T[((lr + 4 * v) << 4) + lc];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int j = j0 + u * NW;
                    if (j < kt) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) Vp[lr + 4 * v][(j << 4) + lc] = (lr + 4 * v < bs) ? x[u][v] : 0.0;
                    }
                }
            }
            if (w == (kt % NW)) {
                KP T = A + tile_base(kt, kt);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int pp = lr + 4 * v;                           // Vp[pp][k0 + lc] = A(k0 + lc, k0 + pp), symmetric
                    const double x = (lc >= pp) ? T[(lc << 4) + pp] : T[(pp << 4) + lc];
                    Vp[pp][k0 + lc] = (pp < bs) ? x : 0.0;
                }
            }
        };
        // (2) wavefront 0: D^-1 of the identity-padded pivot block of panel Vp, Gauss-Jordan in registers -> Pp
        auto invert = [&](double (*Vp)[LDV], double (*Pp)[GP_B], int k0, int bs) {
            if (w != 0) return;
            const int a = l >> 2, c = l & 3;
            double p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = 4 * c + r;
                p[r] = (a < bs && b < bs) ? Vp[b][k0 + a] : ((a == b) ? 1.0 : 0.0);
            }
            bool bad = false;
            gp_inv16_pivots<0>(p, a, c, bad, prod, ld);
#pragma unroll
            for (int r = 0; r < 4; ++r) Pp[a][4 * c + r] = -p[r];         // the sweep leaves -D^-1
            if (bad && l == 0) S.pivot_bad = 1;
        };
        // W_i' = D^-1 V_i'  ->  register v of lane l = W_i(row lc, k = 4 v + lr): the A operand of the updates
        auto w_alayout = [&](const double (&dA)[4], double (*Vp)[LDV], int i) {
            gp_v4f64 wt = {0, 0, 0, 0};
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
                wt = __builtin_amdgcn_mfma_f64_16x16x4f64(dA[kc], Vp[4 * kc + lr][(i << 4) + lc], wt, 0, 0, 0);
            return wt;
        };
        // the same numbers in D (tile) layout: register v of lane l = W_i(row lr + 4 v, col lc)
        auto w_dlayout = [&](const double (&dA)[4], double (*Vp)[LDV], int i) {
            gp_v4f64 wd = {0, 0, 0, 0};
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
                wd = __builtin_amdgcn_mfma_f64_16x16x4f64(Vp[4 * kc + lr][(i << 4) + lc], dA[kc], wd, 0, 0, 0);
            return wd;
        };

        // ---- pivot look-ahead.  The Gauss-Jordan inverse of a 16 x 16 pivot block is a chain of 16 dependent
        // reciprocals on ONE wavefront (19-36 % of an evaluation when every other wavefront waits for it).  The pivot block
        // of the NEXT step is final as soon as the current step has been applied to that one tile, so wavefront 0 applies
        // it to that tile first -- the same MFMAs in the same order as the update loop, which then skips the tile -- and
        // inverts the block while the other wavefronts update the rest of the matrix; the inverse waits in its registers
        // for the next step.  Tile rows are handed out by a ticket (longest first), so wavefront 0 simply takes fewer rows.
        // Operation for operation the arithmetic of the step-by-step schedule: bit-identical results.
        constexpr bool kLookAhead = LDS::kLookAhead;
        double pn1[4] = {0, 0, 0, 0}, pn2[4] = {0, 0, 0, 0};      // wavefront 0: -(-D^-1) of the next pivot block(s), Gauss-Jordan layout
        bool have1 = false, have2 = false;                         // (uniform) the next step's inverse(s) are already known
        auto grab_row = [&]() -> int {
            int r = 0;
            if (l == 0) {
                if constexpr (LDS::kInLds)
                    r = __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)(&S.row_ticket), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else
                    r = __hip_atomic_fetch_add(&S.row_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            return nt - 1 - __builtin_amdgcn_readfirstlane(r);
        };
        // tile in D layout -> Gauss-Jordan layout through the staging tile (lower half mirrored, identity padding beyond
        // bs: exactly what gather + invert read), inverse -> pout (wavefront 0 only)
        auto stage_invert = [&](const gp_v4f64& c, int bs, double (&pout)[4]) {
            const int a = l >> 2, cq = l & 3;
            W::wave_sync();
#pragma unroll
            for (int v = 0; v < 4; ++v) S.la.stage[((lr + 4 * v) << 4) + lc] = c[v];
            W::wave_sync();
            double p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int bq = 4 * cq + r;
                const double x = (a >= bq) ? S.la.stage[(a << 4) + bq] : S.la.stage[(bq << 4) + a];
                p[r] = (a < bs && bq < bs) ? x : ((a == bq) ? 1.0 : 0.0);
            }
            bool bad = false;
            gp_inv16_pivots<0>(p, a, cq, bad, prod, ld);
#pragma unroll
            for (int r = 0; r < 4; ++r) pout[r] = -p[r];
            if (bad && l == 0) S.pivot_bad = 1;
        };
        auto write_P = [&](double (*Pp)[GP_B], const double (&pin)[4]) {
            const int a = l >> 2, cq = l & 3;
#pragma unroll
            for (int r = 0; r < 4; ++r) Pp[a][4 * cq + r] = pin[r];
        };
        // C_ij -= W_i V_j' with W_i in A layout (negated) against panel Vp
        auto rank16 = [&](gp_v4f64 c, const gp_v4f64& wt, double (*Vp)[LDV], int j) {
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-wt[kc], Vp[4 * kc + lr][(j << 4) + lc], c, 0, 0, 0);
            return c;
        };
        auto load_tile = [&](KP T) {
            gp_v4f64 c;
#pragma unroll
            for (int v = 0; v < 4; ++v) c[v] = T[((lr + 4 * v) << 4) + lc];
            return c;
        };
        auto store_tile = [&](KP T, const gp_v4f64& c) {
#pragma unroll
            for (int v = 0; v < 4; ++v) T[((lr + 4 * v) << 4) + lc] = c[v];
        };

        if (W::lane() == 0) S.pivot_bad = 0;
        int kt = 0, k0 = 0;
        while (k0 < n) {
            if constexpr (LDS::kFuse) {
                if (k0 + 2 * GP_B <= n) {
                    // ===== two full pivot tiles a = kt, b = kt + 1 in ONE pass over the matrix (the global-scratch tiers are
                    // bound by the traffic of that pass, not by the MFMAs): step a is first applied to tile column b and tile
                    // row b only, which gives the second panel; then every other tile takes both rank-16 updates at once.
                    // Operation for operation the same arithmetic as two single steps.
                    const int a = kt, b = kt + 1;
                    auto& V2 = S.fz.V2;
                    auto& P2 = S.fz.P2;
                    GP_T0();
                    gather(S.V, a, GP_B);
                    if (w == 0 && have1) write_P(S.P, pn1);
                    if (w == 0 && have2) write_P(P2, pn2);
                    if (W::lane() == 0) S.row_ticket = 0;
                    W::sync();
                    GP_T(0);
                    if (!have1) { invert(S.V, S.P, k0, GP_B); W::sync(); }
                    if (S.pivot_bad != 0) return false;
                    GP_T(1);
                    double dA1[4];
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc) dA1[kc] = S.P[lc][4 * kc + lr];
                    // ---- step a on tile column b (rows i > b, by their owners) ...
                    for (int blk = 0;; ++blk) {
                        const int pos = (blk & 1) ? NW - 1 - w : w;
                        const int i = nt - 1 - (blk * NW + pos);
                        if (i <= b) break;
                        const gp_v4f64 wt = w_alayout(dA1, S.V, i);
                        KP T = A + tile_base(i, b);
                        gp_v4f64 c;
#pragma unroll
                        for (int v = 0; v < 4; ++v) c[v] = T[((lr + 4 * v) << 4) + lc];
#pragma unroll
                        for (int kc = 0; kc < 4; ++kc)
                            c = __builtin_amdgcn_mfma_f64_16x16x4f64(-wt[kc], S.V[4 * kc + lr][(b << 4) + lc], c, 0, 0, 0);
#pragma unroll
                        for (int v = 0; v < 4; ++v) T[((lr + 4 * v) << 4) + lc] = c[v];
                        // the second panel straight from the registers (what gather(V2, b) would read back from global scratch)
#pragma unroll
                        for (int v = 0; v < 4; ++v) V2[lc][(i << 4) + lr + 4 * v] = c[v];
                    }
                    // ---- ... and on tile row b, its b + 1 tiles dealt to all wavefronts
                    {
                        const gp_v4f64 wtb = w_alayout(dA1, S.V, b);
                        for (int j = w; j <= b; j += NW) {
                            KP T = A + tile_base(b, j);
                            gp_v4f64 c;
                            if (j == a) c = w_dlayout(dA1, S.V, b);        // pivot column of step a: A(b-rows, a-cols) <- W_b
                            else {
#pragma unroll
                                for (int v = 0; v < 4; ++v) c[v] = T[((lr + 4 * v) << 4) + lc];
#pragma unroll
                                for (int kc = 0; kc < 4; ++kc)
                                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(-wtb[kc], S.V[4 * kc + lr][(j << 4) + lc], c, 0, 0, 0);
                            }
#pragma unroll
                            for (int v = 0; v < 4; ++v) T[((lr + 4 * v) << 4) + lc] = c[v];
                            if (j < b) {
#pragma unroll
                                for (int v = 0; v < 4; ++v) V2[lr + 4 * v][(j << 4) + lc] = c[v];
                            } else {
                                // the pivot block itself: V2[p][16 b + q] = A(16 b + q, 16 b + p) from the lower triangle, both ways
#pragma unroll
                                for (int v = 0; v < 4; ++v) {
                                    const int pp = lr + 4 * v;
                                    if (lc <= pp) { V2[pp][(b << 4) + lc] = c[v]; V2[lc][(b << 4) + pp] = c[v]; }
                                }
                            }
                        }
                    }
                    W::sync();
                    if (!have2) { invert(V2, P2, k0 + GP_B, GP_B); W::sync(); }
                    if (S.pivot_bad != 0) return false;
                    GP_T(2);
                    double dA2[4];
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc) dA2[kc] = P2[lc][4 * kc + lr];
                    // ---- look-ahead (wavefront 0): the pivot blocks of the next pass, a2 = b + 1 and -- when that pass takes two
                    // tiles again -- b2 = b + 2.  Tiles (a2, a2), (b2, a2) and (b2, b2) get steps a and b here and are skipped below.
                    const int rest = n - (k0 + 2 * GP_B);              // pivots left after this pass
                    const bool la1 = kLookAhead && rest > 0, la2 = kLookAhead && rest >= 2 * GP_B;
                    const int a2 = b + 1, b2 = b + 2;
                    if (la1 && w == 0) {
                        const gp_v4f64 w1a = w_alayout(dA1, S.V, a2), w2a = w_alayout(dA2, V2, a2);
                        KP Taa = A + tile_base(a2, a2);
                        gp_v4f64 c = rank16(rank16(load_tile(Taa), w1a, S.V, a2), w2a, V2, a2);
                        store_tile(Taa, c);
                        stage_invert(c, (rest < GP_B) ? rest : GP_B, pn1);
                        if (la2) {
                            // -D1'^-1 in A-operand layout (through the staging tile)
                            W::wave_sync();
                            write_P(reinterpret_cast<double (*)[GP_B]>(S.la.stage), pn1);
                            W::wave_sync();
                            double dAn[4];
#pragma unroll
                            for (int kc = 0; kc < 4; ++kc) dAn[kc] = S.la.stage[(lc << 4) + 4 * kc + lr];
                            const gp_v4f64 w1b = w_alayout(dA1, S.V, b2), w2b = w_alayout(dA2, V2, b2);
                            KP Tba = A + tile_base(b2, a2);
                            const gp_v4f64 cba = rank16(rank16(load_tile(Tba), w1b, S.V, a2), w2b, V2, a2);
                            store_tile(Tba, cba);
                            // the next pass's first panel at row b2: Vn[p][q] = A(16 b2 + q, 16 a2 + p)
                            W::wave_sync();
#pragma unroll
                            for (int v = 0; v < 4; ++v) S.la.stage[(lc << 4) + lr + 4 * v] = cba[v];
                            W::wave_sync();
                            double vb[4];
#pragma unroll
                            for (int kc = 0; kc < 4; ++kc) vb[kc] = S.la.stage[((4 * kc + lr) << 4) + lc];
                            gp_v4f64 wtn = {0, 0, 0, 0};                 // W'_b2 = (D1'^-1 Vn')' as in w_alayout
#pragma unroll
                            for (int kc = 0; kc < 4; ++kc) wtn = __builtin_amdgcn_mfma_f64_16x16x4f64(dAn[kc], vb[kc], wtn, 0, 0, 0);
                            KP Tbb = A + tile_base(b2, b2);
                            gp_v4f64 cbb = rank16(rank16(load_tile(Tbb), w1b, S.V, b2), w2b, V2, b2);
                            store_tile(Tbb, cbb);                        // steps a and b: what the next pass's step a2 starts from
#pragma unroll
                            for (int kc = 0; kc < 4; ++kc) cbb = __builtin_amdgcn_mfma_f64_16x16x4f64(-wtn[kc], vb[kc], cbb, 0, 0, 0);
                            stage_invert(cbb, GP_B, pn2);
                        }
                    }
                    // ---- both updates on every other tile, row by row
                    for (;;) {
                        const int i = grab_row();
                        if (i < 0) break;
                        if (i == b) {
                            // pivot row of step b: off-diagonal tiles come from the owners of the rows j < b (below)
                            KP T = A + tile_base(b, b);
#pragma unroll
                            for (int v = 0; v < 4; ++v) T[((lr + 4 * v) << 4) + lc] = -P2[lr + 4 * v][lc];
                            continue;
                        }
                        const gp_v4f64 w2 = w_alayout(dA2, V2, i);
                        if (i > b) {
                            KP T = A + tile_base(i, b);                   // A(16 i + lc, 16 b + k) <- W2_i(lc, k)
#pragma unroll
                            for (int v = 0; v < 4; ++v) T[(lc << 4) + 4 * v + lr] = w2[v];
                        } else {
                            KP T = A + tile_base(b, i);                   // A(16 b + k, 16 i + lc) <- W2_i(lc, k)
#pragma unroll
                            for (int v = 0; v < 4; ++v) T[((4 * v + lr) << 4) + lc] = w2[v];
                        }
                        const double n2[4] = {-w2[0], -w2[1], -w2[2], -w2[3]};
                        if (i == a) {
                            // pivot row of step a: tile (a, j) holds W1_j' (-D1^-1 on the diagonal) before step b touches it
                            for (int j = 0; j <= a; ++j) {
                                gp_v4f64 c;
                                if (j == a) {
#pragma unroll
                                    for (int v = 0; v < 4; ++v) c[v] = -S.P[lr + 4 * v][lc];
                                } else c = w_alayout(dA1, S.V, j);       // element (k, col) of the tile = W1_j(col, k)
#pragma unroll
                                for (int kc = 0; kc < 4; ++kc)
                                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(n2[kc], V2[4 * kc + lr][(j << 4) + lc], c, 0, 0, 0);
                                KP T = A + tile_base(a, j);
#pragma unroll
                                for (int v = 0; v < 4; ++v) T[((lr + 4 * v) << 4) + lc] = c[v];
                            }
                            continue;
                        }
                        const gp_v4f64 w1 = w_alayout(dA1, S.V, i);
                        const double n1[4] = {-w1[0], -w1[1], -w1[2], -w1[3]};
                        constexpr int UNR = 2;      // tiles in flight (four were measured slower, also at 256 registers per lane; so was
                                                    // requesting the next trip's tiles ahead of this trip's products: +8 %, 11 more spilled registers)
                        for (int j0 = 0; j0 <= i; j0 += UNR) {
                            gp_v4f64 c[UNR];
                            double b1[UNR][4], b2v[UNR][4];
                            bool on[UNR], col_a[UNR];
#pragma unroll
                            for (int u = 0; u < UNR; ++u) {
                                const int j = j0 + u;
                                // (the look-ahead tiles are wavefront 0's)
                                const bool ahead = (la1 && i == a2 && j == a2) || (la2 && i == b2 && (j == a2 || j == b2));
                                on[u] = (j <= i) && (j != b) && !ahead;
                                col_a[u] = (j == a);
                                const int jj = (j <= i) ? j : i;
                                KP T = A + tile_base(i, jj);
#pragma unroll
                                for (int v = 0; v < 4; ++v) c[u][v] = T[((lr + 4 * v) << 4) + lc];
#pragma unroll
                                for (int kc = 0; kc < 4; ++kc) { b1[u][kc] = S.V[4 * kc + lr][(jj << 4) + lc]; b2v[u][kc] = V2[4 * kc + lr][(jj << 4) + lc]; }
                            }
#pragma unroll
                            for (int u = 0; u < UNR; ++u) {
                                // pivot column of step a: the tile is W1_i itself, not an update of its old content
                                if (col_a[u]) c[u] = w_dlayout(dA1, S.V, i);
                                else {
#pragma unroll
                                    for (int kc = 0; kc < 4; ++kc) c[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(n1[kc], b1[u][kc], c[u], 0, 0, 0);
                                }
                            }
#pragma unroll
                            for (int kc = 0; kc < 4; ++kc)
#pragma unroll
                                for (int u = 0; u < UNR; ++u) c[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(n2[kc], b2v[u][kc], c[u], 0, 0, 0);
#pragma unroll
                            for (int u = 0; u < UNR; ++u) {
                                if (!on[u]) continue;
                                KP T = A + tile_base(i, j0 + u);
#pragma unroll
                                for (int v = 0; v < 4; ++v) T[((lr + 4 * v) << 4) + lc] = c[u][v];
                            }
                        }
                    }
                    W::sync();
                    GP_T(8);
                    have1 = la1;
                    have2 = la2;
                    kt += 2;
                    k0 += 2 * GP_B;
                    continue;
                }
            }
            const int bs = (n - k0 < GP_B) ? n - k0 : GP_B;
            GP_T0();
            gather(S.V, kt, bs);
            if (w == 0 && have1) write_P(S.P, pn1);
            if (W::lane() == 0) S.row_ticket = 0;
            W::sync();
            GP_T(0);
            if (!have1) { invert(S.V, S.P, k0, bs); W::sync(); }
            if (S.pivot_bad != 0) return false;
            GP_T(1);
            // (3) own rows: W_i, pivot column / row tiles, rank-16 update of the other tiles
            double dA[4];
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) dA[kc] = S.P[lc][4 * kc + lr];      // A operand: D^-1 (row lc, k = 4 kc + lr)
            const bool partial = bs < GP_B;
            // look-ahead (wavefront 0): the next pivot block, tile (kt + 1, kt + 1) after this step
            const int rest = n - (k0 + GP_B);
            const bool la = kLookAhead && rest > 0;
            if (la && w == 0) {
                const gp_v4f64 wn = w_alayout(dA, S.V, kt + 1);
                KP Tn = A + tile_base(kt + 1, kt + 1);
                const gp_v4f64 c = rank16(load_tile(Tn), wn, S.V, kt + 1);
                store_tile(Tn, c);
                stage_invert(c, (rest < GP_B) ? rest : GP_B, pn1);
            }
            for (;;) {
                const int i = grab_row();
                if (i < 0) break;
                if (i == kt && !partial) {
                    // the whole tile row is pivot rows: its off-diagonal tiles are written by the owners of
                    // the rows j < kt (below); the diagonal tile becomes -D^-1
                    KP T = A + tile_base(kt, kt);
#pragma unroll
                    for (int v = 0; v < 4; ++v) T[((lr + 4 * v) << 4) + lc] = -S.P[lr + 4 * v][lc];
                    continue;
                }
                const gp_v4f64 wt = w_alayout(dA, S.V, i);
                if (i > kt) {
                    KP T = A + tile_base(i, kt);                          // A(16 i + lc, k0 + k) <- W_i(lc, k)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (4 * v + lr < bs) T[(lc << 4) + 4 * v + lr] = wt[v];
                } else if (i < kt) {
                    KP T = A + tile_base(kt, i);                          // A(k0 + k, 16 i + lc) <- W_i(lc, k)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (4 * v + lr < bs) T[((4 * v + lr) << 4) + lc] = wt[v];
                } else {
                    // partial last block: rows lc >= bs of the pivot tile row are ordinary rows (the augmented one)
                    KP T = A + tile_base(kt, kt);
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int k = 4 * v + lr;
                        if (k < bs) T[(lc << 4) + k] = (lc >= bs) ? wt[v] : -S.P[lc][k];
                    }
                }
                const double na[4] = {-wt[0], -wt[1], -wt[2], -wt[3]};
                constexpr int UNR = 4;                                     // tiles in flight (independent accumulators)
                for (int j0 = 0; j0 <= i; j0 += UNR) {
                    gp_v4f64 c[UNR];
                    double bv[UNR][4];
                    bool on[UNR];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const int j = j0 + u;
                        on[u] = (j <= i) && (j != kt || i == kt) && !(la && i == kt + 1 && j == kt + 1);
                        const int jj = (j <= i) ? j : i;
                        KP T = A + tile_base(i, jj);
#pragma unroll
                        for (int v = 0; v < 4; ++v) c[u][v] = T[((lr + 4 * v) << 4) + lc];
#pragma unroll
                        for (int kc = 0; kc < 4; ++kc) bv[u][kc] = S.V[4 * kc + lr][(jj << 4) + lc];
                    }
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc)
#pragma unroll
                        for (int u = 0; u < UNR; ++u) c[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(na[kc], bv[u][kc], c[u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        if (!on[u]) continue;
                        const int j = j0 + u;
                        KP T = A + tile_base(i, j);
                        if (i != kt) {
#pragma unroll
                            for (int v = 0; v < 4; ++v) T[((lr + 4 * v) << 4) + lc] = c[u][v];
                        } else {
                            // partial pivot tile row: only its non-pivot rows (and, in the diagonal tile, columns) move
#pragma unroll
                            for (int v = 0; v < 4; ++v)
                                if (lr + 4 * v >= bs && (j != kt || lc >= bs)) T[((lr + 4 * v) << 4) + lc] = c[u][v];
                        }
                    }
                }
            }
            W::sync();
            GP_T(8);
            have1 = la;
            have2 = false;
            ++kt;
            k0 += GP_B;
        }
        // log|K| = (sum of exponents) ln 2 + log(mantissa product), known to wavefront 0
        double tot = (w == 0) ? (ld * 0.6931471805599453 + log(prod)) : 0.0;
        logdet = W::bcast_from_first_wave(tot);
        return true;
    }
#endif
    // ---- host simulation (one lane): the same blocked sweep in plain loops
    double ld = 0.0;
    static thread_local double Wm[GP_B][NP + 16];
    for (int k0 = 0; k0 < n; k0 += GP_B) {
        const int bs = (n - k0 < GP_B) ? n - k0 : GP_B;
        for (int p = 0; p < GP_B; ++p)
            for (int i = 0; i < nrow; ++i) {
                const int kp = k0 + p;
                S.V[p][i] = (p < bs) ? ((i >= kp) ? A[tri_index(i, kp)] : A[tri_index(kp, i)]) : 0.0;
            }
        double (*Pw)[GP_B] = S.P;
        for (int a = 0; a < GP_B; ++a)
            for (int b = 0; b < GP_B; ++b) Pw[a][b] = (a < bs && b < bs) ? S.V[b][k0 + a] : ((a == b) ? 1.0 : 0.0);
        for (int q = 0; q < GP_B; ++q) {
            const double d = Pw[q][q];
            if (!(d > 0.0)) return false;
            ld += log(d);
            const double inv = 1.0 / d;
            double nv[GP_B][GP_B];
            for (int a = 0; a < GP_B; ++a)
                for (int b = 0; b < GP_B; ++b) {
                    double v = Pw[a][b] - Pw[a][q] * (Pw[q][b] * inv);
                    if (a == q) v = Pw[q][b] * inv;
                    if (b == q) v = (a == q) ? -inv : Pw[a][q] * inv;
                    nv[a][b] = v;
                }
            for (int a = 0; a < GP_B; ++a)
                for (int b = 0; b < GP_B; ++b) Pw[a][b] = nv[a][b];
        }
        for (int a = 0; a < GP_B; ++a)
            for (int b = 0; b < GP_B; ++b) Pw[a][b] = -Pw[a][b];       // D^-1 (identity-padded)
        for (int i = 0; i < nrow; ++i)
            for (int p = 0; p < GP_B; ++p) {
                double sacc = 0;
                for (int q = 0; q < GP_B; ++q) sacc = fma(S.V[q][i], Pw[q][p], sacc);
                Wm[p][i] = sacc;
            }
        for (int i = 0; i < nrow; ++i) {
            const bool row_in = (i >= k0 && i < k0 + bs);
            for (int j = 0; j <= i; ++j) {
                const bool col_in = (j >= k0 && j < k0 + bs);
                if (row_in && col_in) A[tri_index(i, j)] = -Pw[i - k0][j - k0];
                else if (col_in) A[tri_index(i, j)] = Wm[j - k0][i];
                else if (row_in) A[tri_index(i, j)] = Wm[i - k0][j];
                else {
                    double acc = 0;
                    for (int p = 0; p < GP_B; ++p) acc = fma(Wm[p][i], S.V[p][j], acc);
                    A[tri_index(i, j)] -= acc;
                }
            }
        }
    }
    (void)nt;
    logdet = ld;
    return true;
}

// Matern-3/2 kernel value and the common derivative factor e = 1.5*c*exp(-u)
LCFE_FN double gp_kernel(double dt2, double dl2, double c, double m0, double m1, double& e) {
    const double u = sqrt(3.0 * (dt2 / m0 + dl2 / m1));
    const double ex = exp(-u);
    e = 1.5 * c * ex;
    return c * (1.0 + u) * ex;
}

// Device tile passes: the same kernel with the two metric divisions replaced by multiplications with 1/M0, 1/M1
// (computed once per evaluation) -- q0 = dt^2/M0 and q1 = dl^2/M1 are also what the gradient needs.
LCFE_FN double gp_kernel_q(double q0, double q1, double c, double& e) {
    const double u = sqrt(3.0 * (q0 + q1));
    const double ex = exp(-u);
    e = 1.5 * c * ex;
    return c * (1.0 + u) * ex;
}

// One evaluation of f = -log-likelihood and its gradient at p (george GP.log_likelihood /
// grad_log_likelihood as wrapped by multiband_gp.py:141-154).  K is overwritten (by -K^-1).  On a
// failed factorisation f = 1e25 and g = 0.  `need_grad` false: only alpha and f (prediction pass).
template <class W, int NP, class KP, class LDS>
LCFE_FN_NOINLINE void gp_eval(const double* p, int n, LDS& S, KP K, double& f, double* g, bool need_grad) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (LDS::kInLds) __builtin_assume(__builtin_amdgcn_is_shared((const void*)&S));    // ds_* addressing of the working set
#endif
    const int lane = W::lane();
    constexpr int G = (W::LANES >= 64) ? 64 : W::LANES;
    constexpr int RG = W::LANES / G;
    const int rl = lane / G, cl = lane % G;
    const double mu = p[0], c = exp(p[1]), m0 = exp(p[2]), m1 = exp(p[3]);
    const double im0 = 1.0 / m0, im1 = 1.0 / m1;
    (void)im0; (void)im1;
    GP_T0();
    // Gram matrix, tile-packed lower triangle
#if defined(__HIPCC__)
    if constexpr (W::WAVE == 64) {
        // every wavefront fills the tiles of the rows it owns in the sweep (D-operand layout: lane l, register v
        // = element (lr + 4v, lc) of the tile): no index arithmetic per element, conflict-free 512-byte stores
        constexpr int NW = W::NWAVES;
        const int l = W::wlane(), lr = l >> 4, lc = l & 15, w = W::wave_id();
        const int nt = (n + 16) >> 4;
        for (int blk = 0;; ++blk) {
            const int pos = (blk & 1) ? NW - 1 - w : w;
            const int i = nt - 1 - (blk * NW + pos);
            if (i < 0) break;
            double ti[4], li[4], ni[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = (i << 4) + lr + 4 * v;
                ti[v] = S.t[r]; li[v] = S.sm.lam_at(r); ni[v] = S.e2[r] + GP_TINY;
            }
            for (int j = 0; j <= i; ++j) {
                const int cj = (j << 4) + lc;
                const double tj = S.t[cj], lj = S.sm.lam_at(cj), rj = S.y[cj] - mu;
                KP T = K + tile_base(i, j);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int r = (i << 4) + lr + 4 * v;
                    const double dt = ti[v] - tj, dl = li[v] - lj;
                    double e;
                    double k = gp_kernel_q(dt * dt * im0, dl * dl * im1, c, e);
                    if (r == cj) k += ni[v];
                    // rows / columns beyond the points: the augmented residual row, zeros elsewhere
                    if (r >= n || cj >= n) k = (r == n && cj < n) ? rj : 0.0;
                    T[((lr + 4 * v) << 4) + lc] = k;
                }
            }
        }
    } else
#endif
    {
    for (int i = rl; i < n; i += RG) {
        const double ti = S.t[i], li = S.sm.lam_at(i);
        for (int j = cl; j <= i; j += G) {
            const double dt = ti - S.t[j], dl = li - S.sm.lam_at(j);
            double e;
            double k = gp_kernel(dt * dt, dl * dl, c, m0, m1, e);
            if (j == i) k += S.e2[i] + GP_TINY;
            K[tri_index(i, j)] = k;
        }
    }
    for (int i = lane; i <= n; i += W::LANES) K[tri_index(n, i)] = (i < n) ? S.y[i] - mu : 0.0;   // augmented row
    }
    W::sync();
    GP_T(4);
    double logdet;
    g[0] = g[1] = g[2] = g[3] = 0.0;
    if (!gp_sweep_inverse<W, NP, KP, LDS>(K, n, S, logdet)) { f = 1e25; W::sync(); return; }
    GP_T(5);      // (sweep total)
    // alpha = the swept augmented row ; r' K^-1 r = -A(n, n)
    double sa = 0;
    for (int i = lane; i < n; i += W::LANES) {
        const double v = K[tri_index(n, i)];
        S.alpha[i] = v;
        sa += v;
    }
    sa = W::sum(sa);
    const double ra = -K[tri_index(n, n)];
    const double ll = -0.5 * (ra + logdet + n * GP_LOG_2PI);
    f = finite_d(ll) ? -ll : 1e25;
    W::sync();
    GP_T(6);
    if (!need_grad) return;
    // gradient: 0.5 * sum_ij (alpha_i alpha_j - Kinv_ij) dK_ij/dtheta ,  Kinv_ij = -K[ij]
    double g1 = 0, g2 = 0, g3 = 0;
#if defined(__HIPCC__)
    if constexpr (W::WAVE == 64) {
        constexpr int NW = W::NWAVES;
        const int l = W::wlane(), lr = l >> 4, lc = l & 15, w = W::wave_id();
        const int nt = (n + 15) >> 4;                    // tile rows that hold points
        for (int blk = 0;; ++blk) {
            const int pos = (blk & 1) ? NW - 1 - w : w;
            const int i = nt - 1 - (blk * NW + pos);
            if (i < 0) break;
            double ti[4], li[4], ai[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = (i << 4) + lr + 4 * v;
                ti[v] = S.t[r]; li[v] = S.sm.lam_at(r); ai[v] = S.alpha[r];
            }
            for (int j = 0; j <= i; ++j) {
                const int cj = (j << 4) + lc;
                const double tj = S.t[cj], lj = S.sm.lam_at(cj), aj = S.alpha[cj];
                KP T = K + tile_base(i, j);
                // the tile's four entries of this lane in ONE round of loads ahead of the arithmetic (every tile (i, j <= i) has
                // storage; a load inside the `in` arm below becomes a branch with its own wait per entry)
                double tv[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) tv[v] = T[((lr + 4 * v) << 4) + lc];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int r = (i << 4) + lr + 4 * v;
                    // lower triangle inside the points: weight 2 off the diagonal, 1 on it; everything else is
                    // masked out (its inputs may be stale memory)
                    const bool in = (r < n && cj <= r);
                    const double dt = in ? ti[v] - tj : 0.0, dl = in ? li[v] - lj : 0.0;
                    const double q0 = dt * dt * im0, q1 = dl * dl * im1;
                    double e;
                    const double k = gp_kernel_q(q0, q1, c, e);
                    const double a_in = (ai[v] * aj + tv[v]) * ((cj == r) ? 1.0 : 2.0);
                    const double a = in ? a_in : 0.0;
                    g1 += a * k;
                    g2 += a * e * q0;
                    g3 += a * e * q1;
                }
            }
        }
    } else
#endif
    for (int i = rl; i < n; i += RG) {
        const double ai = S.alpha[i], ti = S.t[i], li = S.sm.lam_at(i);
        for (int j = cl; j <= i; j += G) {
            const double dt = ti - S.t[j], dl = li - S.sm.lam_at(j);
            const double dt2 = dt * dt, dl2 = dl * dl;
            double e;
            const double k = gp_kernel(dt2, dl2, c, m0, m1, e);
            const double a = (ai * S.alpha[j] + K[tri_index(i, j)]) * ((j == i) ? 1.0 : 2.0);
            g1 += a * k;
            g2 += a * e * dt2 / m0;
            g3 += a * e * dl2 / m1;
        }
    }
    g1 = W::sum(g1);
    g2 = W::sum(g2);
    g3 = W::sum(g3);
    // objective is the NEGATIVE log-likelihood
    g[0] = -sa;
    g[1] = -0.5 * g1;
    g[2] = -0.5 * g2;
    g[3] = -0.5 * g3;
    W::sync();
    GP_T(7);
}

LCFE_FN bool gp_row_valid(const ObjIn& in, int i) {
    // multiband_gp.py:51-59: known band, flux and error not NaN, error > 0
    const double f = in.f[i], e = in.e[i];
    return (in.b[i] < 6) && !is_nan(f) && !is_nan(e) && (e > 0);
}

// multiband_gp.py:292-344 for one object (rows read straight from the CSR slice, file order).
// `K` points to packed-triangle storage for NP points (LDS or global scratch).
// `ev(p, n, f, g, need_grad)` evaluates the objective (gp_eval with the matrix in LDS / global
// scratch, or gp_eval_reg with the matrix in registers).
template <class W, int NP, class Ev, class LDS>
LCFE_FN void gp_object(const ObjIn& L, LDS& S, Ev&& gp_ev, int32_t* st) {
    const int lane = W::lane();
    double* o = S.out;
    for (int k = lane; k < GP_NCOL; k += W::LANES) o[k] = qnan();
#ifdef LCFE_GP_PROF
    if (lane == 0) for (int k = 0; k < 12; ++k) S.prof[k] = 0;
#endif
    // ---- prepare_multiband_data (:34-87): valid rows in file order
    int n = 0;
    double tmin_all = __builtin_inf();
    for (int i = lane; i < L.n; i += W::LANES) {
        tmin_all = fmin(tmin_all, L.t[i]);
        n += gp_row_valid(L, i) ? 1 : 0;
    }
    n = W::sum(n);
    if (n >= 10 && n + 1 <= NP) {
        for (int i = lane; i < L.n; i += W::LANES) {
            if (!gp_row_valid(L, i)) continue;
            int pos = 0;
            for (int j = 0; j < i; ++j) pos += gp_row_valid(L, j) ? 1 : 0;
            S.t[pos] = L.t[i];
            S.sm.set_band(pos, L.b[i]);
            S.y[pos] = L.f[i];
            S.e2[pos] = L.e[i];
        }
    }
    tmin_all = W::min(tmin_all);
    W::sync();
    if (st && lane == 0) { st[0] = 0; st[1] = 0; st[2] = 0; st[3] = n; }
    if (n < 10) { W::sync(); return; }                                  // :66 -> all 27 NaN
    if (n + 1 > NP) { if (st && lane == 0) st[0] = -100; W::sync(); return; }   // + the augmented row of gp_eval_reg
    double tmin = __builtin_inf();
    int nnz = 0;
    for (int i = lane; i < n; i += W::LANES) { tmin = fmin(tmin, S.t[i]); }
    tmin = W::min(tmin);
    // flux_scale = median(|f| over f != 0) (:78-80): zeros are parked at +inf so that the non-zero
    // values occupy ranks 0..nnz-1 (scratch vector: see GpSmallArrays)
    double* rscr = S.sm.rbuf(S.V);
    for (int i = lane; i < n; i += W::LANES) {
        const bool nz = (S.y[i] != 0.0);
        rscr[i] = nz ? fabs(S.y[i]) : __builtin_inf();
        nnz += nz ? 1 : 0;
    }
    nnz = W::sum(nnz);
    W::sync();
    double scale = qnan();
    if (nnz > 0) {
        wave_rank_select<W>(rscr, n, (nnz - 1) / 2, nnz / 2, S.slot, reinterpret_cast<unsigned long long*>(S.alpha));
        scale = ((nnz & 1) ? S.slot[0] : (S.slot[0] + S.slot[1]) / 2.0);
        W::sync();
    }
    if (scale == 0) scale = 1.0;
    double sy = 0;
    for (int i = lane; i < n; i += W::LANES) {
        S.t[i] -= tmin;                                                 // :75
        S.y[i] /= scale;                                                // :81-82
        const double e = S.e2[i] / scale;
        S.e2[i] = e * e;
        sy += S.y[i];
    }
    W::sync();
    const double ymean = W::sum(sy) / n;
    double sv = 0;
    for (int i = lane; i < n; i += W::LANES) { const double d = S.y[i] - ymean; sv += d * d; }
    const double yvar = W::sum(sv) / n;                                 // :125 np.var
    // ---- fit_multiband_gp (:90-193)
    double p[4] = {ymean, log(yvar / 2.0), log(100.0 * 100.0), log(6000.0 * 6000.0)};   // amp/ndim: george `float * kernel`
    bool finite0 = finite_d(p[0]) && finite_d(p[1]);
    double fval = 0;
    double xe[4] = {qnan(), qnan(), qnan(), qnan()}, fe_last = 1e25;     // the point of the optimiser's last evaluation, its objective
    int n_iter = 0, n_eval = 0, why = LB_ERROR;
    if (finite0) {
        // the optimiser's state machine lives in LDS and is advanced by ONE thread between two barriers:
        // none of its state is live in registers while the objective (the wide code) runs
        if (lane == 0) lb_start(S.lb, p, 100, 1e7, 1e-5, 20);
        W::sync();
        for (;;) {
            double fe, ge[4];
            xe[0] = S.lb.x[0]; xe[1] = S.lb.x[1]; xe[2] = S.lb.x[2]; xe[3] = S.lb.x[3];
            gp_ev(S.lb.x, n, fe, ge, true);
            fe_last = fe;
            if (lane == 0) {
                S.lb.f = fe;
                S.lb.g[0] = ge[0]; S.lb.g[1] = ge[1]; S.lb.g[2] = ge[2]; S.lb.g[3] = ge[3];
#if defined(LCFE_GP_PROF) && defined(__HIPCC__)
                const unsigned long long tlb = __builtin_readcyclecounter();
#endif
                S.lb_why = lb_advance<4, 10, LDS::kInLds>(S.lb);
#if defined(LCFE_GP_PROF) && defined(__HIPCC__)
                S.prof[9] += __builtin_readcyclecounter() - tlb;
#endif
            }
            W::sync();
            if (S.lb_why != LB_EVAL) break;
        }
        why = S.lb_why;
        n_iter = S.lb.n_iter;
        n_eval = S.lb.n_eval;
        fval = S.lb.f;
        p[0] = S.lb.x[0]; p[1] = S.lb.x[1]; p[2] = S.lb.x[2]; p[3] = S.lb.x[3];
    }
    if (st && lane == 0) { st[0] = why; st[1] = n_iter; st[2] = n_eval; }
#ifdef LCFE_GP_PROF
    if (st && lane == 0) for (int k = 0; k < 10; ++k) st[4 + k] = (int)(S.prof[k] >> 10);
#endif
    if (!finite0 || !(finite_d(p[0]) && finite_d(p[1]) && finite_d(p[2]) && finite_d(p[3]))) { W::sync(); return; }
    // features read params[0..2] of george's vector [mean, log_constant, log_M_0_0, log_M_1_1] (:171-188)
    const double amplitude = exp(p[0]);
    const double ts = sqrt(exp(p[1]));
    const double ws = sqrt(exp(p[2]));
    if (lane == 0) {
        o[0] = amplitude; o[1] = ts; o[2] = ws; o[3] = -fval; o[4] = ts / (ws / 1000);
    }
    // ---- peak time (:331-338): first max of the r rows (file order; pandas idxmax skips NaN), else of all rows
    int pk = -1;
    const ObjIn& Lr = L;
    (void)Lr;
    {
        double best = -__builtin_inf();
        int nr = 0;
        for (int i = lane; i < L.n; i += W::LANES) nr += (L.b[i] == 2);
        nr = W::sum(nr);
        const bool use_r = nr > 0;
        for (int i = lane; i < L.n; i += W::LANES)
            if ((!use_r || L.b[i] == 2) && !is_nan(L.f[i])) best = fmax(best, L.f[i]);
        best = W::max(best);
        int cand = 0x7fffffff;
        for (int i = lane; i < L.n; i += W::LANES)
            if ((!use_r || L.b[i] == 2) && L.f[i] == best) cand = (i < cand) ? i : cand;
        pk = W::min(cand);
    }
    if (pk == 0x7fffffff) { W::sync(); return; }
    const double peak_time = L.t[pk] - tmin_all;
    // ---- interpolate_multiband (:196-289): alpha at the optimum, then 12 predictions.  L-BFGS-B normally ends ON the
    // point it evaluated last (the accepted end of its line search): S.alpha then already is K^-1 r at the optimum -- the
    // same code computed it from the same inputs -- and the extra factorisation (one sweep in 26) is only paid when the
    // optimiser stepped back to an earlier iterate.
    double ftmp = fe_last, gtmp[4];
    if (!(p[0] == xe[0] && p[1] == xe[1] && p[2] == xe[2] && p[3] == xe[3])) gp_ev(p, n, ftmp, gtmp, false);
    if (ftmp >= 1e25) { W::sync(); return; }          // factorisation failed at the optimum: predict raises -> NaN (:279-287)
    const double c = exp(p[1]), m0 = exp(p[2]), m1 = exp(p[3]);
    const double EP[4] = {0, 20, 50, 100};
    const int PB[3] = {1, 2, 3};
    double fl[12];
    for (int q = 0; q < 12; ++q) {
        const double tp = peak_time + EP[q / 3], lp = gp_wavelength(PB[q % 3]);
        double s = 0;
        for (int i = lane; i < n; i += W::LANES) {
            const double dt = tp - S.t[i], dl = lp - S.sm.lam_at(i);
            double e;
            s += gp_kernel(dt * dt, dl * dl, c, m0, m1, e) * S.alpha[i];
        }
        fl[q] = (p[0] + W::sum(s)) * scale;                              // :244-247
    }
    if (lane == 0) {
        double gr[4];
        for (int e = 0; e < 4; ++e) {
            const double gf = fl[3 * e], rf = fl[3 * e + 1], iff = fl[3 * e + 2];
            o[5 + 5 * e] = gf; o[6 + 5 * e] = rf; o[7 + 5 * e] = iff;
            gr[e] = (gf > 0 && rf > 0) ? -2.5 * log10(gf / rf) : qnan();             // :254-262
            o[8 + 5 * e] = gr[e];
            o[9 + 5 * e] = (rf > 0 && iff > 0) ? -2.5 * log10(rf / iff) : qnan();
        }
        o[25] = (!is_nan(gr[0]) && !is_nan(gr[2])) ? (gr[2] - gr[0]) / 50.0 : qnan();   // :265-277
        o[26] = (!is_nan(gr[0]) && !is_nan(gr[3])) ? (gr[3] - gr[0]) / 100.0 : qnan();
    }
    W::sync();
}

}  // namespace lcfe
