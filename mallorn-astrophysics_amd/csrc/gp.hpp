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
#endif

template <class P> struct gp_is_lds_ptr { static constexpr bool value = false; };
#if defined(__HIPCC__)
template <> struct gp_is_lds_ptr<lds_double*> { static constexpr bool value = true; };
#endif

// pivot-block width of the sweep: 8 for the LDS tiers and the 768-point fallback, 16 for the
// 512-point global-scratch tier (half as many block steps, each with four MFMAs per tile)
template <int NP> struct gp_block { static constexpr int B = (NP == 512 || NP == 240) ? 16 : 8; };

// Working memory of one object; NP = capacity in points.  The packed matrix itself (`K`, NP(NP+1)/2
// doubles) lives in LDS for the small tiers and in a per-workgroup slab of global scratch otherwise.
template <int NP, int NW = 4>
struct GpLds {
    double t[NP], lam[NP], y[NP], e2[NP];     // valid points: time (from first valid), wavelength, flux/scale, (err/scale)^2
    double r[NP], alpha[NP];                  // residual y - mu ; K^-1 r
    double V[gp_block<NP>::B][NP];            // pivot-block columns A(:, P)
    double P[(NW <= 4) ? NW : 1][gp_block<NP>::B][gp_block<NP>::B];  // per-wavefront copy (one shared copy for > 4 waves) of the pivot block -> minus its inverse
    double Wm[gp_block<NP>::B][NP];           // A(:, P) * A(P,P)^-1
    double lb_s[10][4], lb_y[10][4], lb_rho[10];   // L-BFGS memory (block-uniform, kept out of registers)
    double slot[2];
    double out[GP_NCOL + 1];
    int pivot_bad;                            // shared-copy mode: wave 0 reports a non-positive pivot
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

LCFE_FN double gp_wavelength(int band) {
    // multiband_gp.py:26-29
    const double w[6] = {3670.0, 4825.0, 6222.0, 7545.0, 8691.0, 9710.0};
    return w[band];
}

#if defined(__HIPCC__)
typedef double gp_v4f64 __attribute__((ext_vector_type(4)));
#endif

// Rank-8 update of all tiles for one pivot block.  On the GPU each wavefront takes whole tiles and
// does the 16 x 16 x 8 product with two v_mfma_f64_16x16x4_f64 (operand layout measured on gfx950,
// tools/mfma_f64_probe.hip: A lane l = (row l%16, k l/16), B lane l = (k l/16, col l%16), D lane l,
// register v = (row l/16 + 4v, col l%16)); the host simulation uses plain loops.
template <class W, int NP, class KP>
LCFE_FN void gp_tile_update(KP A, int n, GpLds<NP, W::NWAVES>& S, int k0, int bs) {   // n = rows incl. the augmented one
    constexpr int B = gp_block<NP>::B;
    constexpr int KC = B / 4;                   // 16x16x4 MFMAs per tile
    (void)KC;
    const int nt = (n + 15) >> 4;
#if defined(__HIPCC__)
    if constexpr (W::WAVE == 64) {
        const int l = W::wlane();
        const int lr = l >> 4, lc = l & 15;
        const int ntile = nt * (nt + 1) / 2;
        constexpr int UNR = 4;                 // tiles in flight per wavefront (independent register sets)
        for (int t0 = W::wave_id() * UNR; t0 < ntile; t0 += W::NWAVES * UNR) {
            int ti[UNR], tj[UNR];
            bool on[UNR];
            gp_v4f64 c[UNR];
            double av[UNR][KC], bv[UNR][KC];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int t = t0 + u;
                on[u] = t < ntile;
                const int tt = on[u] ? t : 0;
                int r = (int)((sqrtf(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f);      // row of the linear tile index
                while (r * (r + 1) / 2 > tt) --r;
                while ((r + 1) * (r + 2) / 2 <= tt) ++r;
                ti[u] = r;
                tj[u] = tt - r * (r + 1) / 2;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                KP base = A + tile_base(ti[u], tj[u]);
#pragma unroll
                for (int v = 0; v < 4; ++v) c[u][v] = base[((lr + 4 * v) << 4) + lc];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int ri = (ti[u] << 4) + lc, cj = (tj[u] << 4) + lc;   // operand row / column of this lane
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) { av[u][kc] = -S.Wm[4 * kc + lr][ri]; bv[u][kc] = S.V[4 * kc + lr][cj]; }
            }
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                for (int u = 0; u < UNR; ++u) c[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][kc], bv[u][kc], c[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (!on[u]) continue;
                KP base = A + tile_base(ti[u], tj[u]);
                const int col = (tj[u] << 4) + lc;
                const bool col_in = (col >= k0 && col < k0 + bs);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = (ti[u] << 4) + lr + 4 * v;
                    const bool row_in = (row >= k0 && row < k0 + bs);
                    if (!col_in && !row_in) base[((lr + 4 * v) << 4) + lc] = c[u][v];
                }
            }
        }
        return;
    }
#endif
    (void)nt;
    for (int i = W::lane(); i < n; i += W::LANES) {
        if (i >= k0 && i < k0 + bs) continue;
        for (int j = 0; j <= i; ++j) {
            if (j >= k0 && j < k0 + bs) continue;
            double acc = 0;
            for (int p = 0; p < gp_block<NP>::B; ++p) acc = fma(S.Wm[p][i], S.V[p][j], acc);
            A[tri_index(i, j)] -= acc;
        }
    }
}

// Pw <- -(identity-padded pivot block)^-1 by B single-index sweeps; returns true on a non-positive
// pivot and adds log(pivots) to ld.  On the GPU the 8 x 8 block lives in the registers of one
// wavefront (lane e holds element (e/8, e%8)) and the pivot row / column are fetched with lane
// shuffles -- no LDS round trip per pivot; the host simulation goes through the Pw array.
template <class W, int NP>
LCFE_FN bool gp_pivot_inverse(GpLds<NP, W::NWAVES>& S, double (*Pw)[gp_block<NP>::B], int k0, int bs, double& ld) {
    constexpr int B = gp_block<NP>::B;
    bool bad = false;
    double prod = 1.0;
#if defined(__HIPCC__)
    if constexpr (W::WAVE == 64 && B == 8) {
        const int e = W::wlane();
        const int a = e >> 3, b = e & 7;
        double p = (a < bs && b < bs) ? S.V[b][k0 + a] : ((a == b) ? 1.0 : 0.0);
#pragma unroll
        for (int q = 0; q < B; ++q) {
            const double d = __shfl(p, q * 9, 64);                 // pivot (q, q): uniform
            const double paq = __shfl(p, (e & ~7) | q, 64);        // (a, q)
            const double pqb = __shfl(p, (q << 3) | b, 64);        // (q, b)
            if (!(d > 0.0)) bad = true;
            prod *= d;
            const double inv = 1.0 / d;
            double v = p - paq * pqb * inv;
            if (a == q || b == q) v = ((a == q) ? pqb : paq) * inv;
            if (a == q && b == q) v = -inv;
            p = v;
        }
        ld += log(prod);
        Pw[a][b] = p;
        W::wave_sync();
        return bad;
    }
#endif
    for (int e = W::wlane(); e < B * B; e += W::WAVE) {
        const int a = e / B, b = e % B;
        Pw[a][b] = (a < bs && b < bs) ? S.V[b][k0 + a] : ((a == b) ? 1.0 : 0.0);
    }
    W::wave_sync();
    for (int q = 0; q < B; ++q) {
        const double d = Pw[q][q];
        if (!(d > 0.0)) bad = true;
        prod *= d;
        if ((q & 7) == 7) { ld += log(prod); prod = 1.0; }          // one log per 8 pivots (no overflow)
        const double inv = 1.0 / d;
        constexpr int NV = (B * B + W::WAVE - 1) / W::WAVE;
        double nv[NV];
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int e = W::wlane() + c * W::WAVE;
            const int a = e / B, b = e % B;
            const double paq = Pw[a][q], pqb = Pw[q][b], pab = Pw[a][b];
            double v = pab - paq * pqb * inv;
            if (a == q || b == q) v = ((a == q) ? pqb : paq) * inv;
            if (a == q && b == q) v = -inv;
            nv[c] = v;
        }
        W::wave_sync();
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int e = W::wlane() + c * W::WAVE;
            Pw[e / B][e % B] = nv[c];
        }
        W::wave_sync();
    }
    return bad;
}

// Wm = V^T * (-Pw): 16 rows of the matrix per MFMA pair on the GPU (B operand = the 8 x 8 block
// padded to 16 columns), one row per lane otherwise.
template <class W, int NP>
LCFE_FN void gp_row_weights(GpLds<NP, W::NWAVES>& S, const double (*Pw)[gp_block<NP>::B], int n) {
    constexpr int B = gp_block<NP>::B;
#if defined(__HIPCC__)
    if constexpr (W::WAVE == 64) {
        const int l = W::wlane();
        const int lr = l >> 4, lc = l & 15;
        // B operand: (k = 4 kc + lr, col = lc): -Pw[k][col] for col < B, zero padding beyond
        constexpr int KC = B / 4;
        double bq[KC];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) bq[kc] = (lc < B) ? -Pw[4 * kc + lr][lc] : 0.0;
        const int nt = (n + 15) >> 4;
        for (int t = W::wave_id(); t < nt; t += W::NWAVES) {
            const int ri = (t << 4) + lc;
            gp_v4f64 c = {0, 0, 0, 0};
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)                                   // A operand: (row = lc, k = 4 kc + lr)
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(S.V[4 * kc + lr][ri], bq[kc], c, 0, 0, 0);
            // D lane (row = lr + 4v, col = lc): Wm[col][row]
            if (lc < B) {
#pragma unroll
                for (int v = 0; v < 4; ++v) S.Wm[lc][(t << 4) + lr + 4 * v] = c[v];
            }
        }
        return;
    }
#endif
    for (int i = W::lane(); i < n; i += W::LANES) {
        double vq[B];
#pragma unroll
        for (int q = 0; q < B; ++q) vq[q] = S.V[q][i];
#pragma unroll
        for (int p = 0; p < B; ++p) {
            double sacc = 0;
#pragma unroll
            for (int q = 0; q < B; ++q) sacc = fma(-vq[q], Pw[q][p], sacc);
            S.Wm[p][i] = sacc;
        }
    }
}

// In-place inverse of the packed symmetric positive-definite matrix A (lower triangle, row-major)
// by a BLOCKED SYMMETRIC SWEEP: for each pivot block P of B consecutive indices
//     A_PP <- -A_PP^-1 ,  A_RP <- A_RP A_PP^-1 ,  A_RR <- A_RR - A_RP A_PP^-1 A_PR     (R = all other indices)
// After all blocks A = -K^-1.  The pivots met while inverting the diagonal blocks are exactly the
// Cholesky pivots L_jj^2, so log|K| = sum log(pivot) and "pivot <= 0" is LAPACK dpotrf's failure
// (george: log-likelihood = -inf).  n/B block steps with two workgroup barriers each; a step is a
// fully parallel rank-B update of the whole triangle -- this replaces the n sequential columns of
// a textbook Cholesky + triangular inverse.
// The matrix carries one extra, never pivoted row n (the residual r = y - mu): the sweep turns it
// into alpha = K^-1 r and its diagonal into -r'K^-1 r (regression use of the sweep operator), so
// no solve or matrix-vector product is needed afterwards.
template <class W, int NP, class KP>
LCFE_FN bool gp_sweep_inverse(KP A, int n, GpLds<NP, W::NWAVES>& S, double& logdet) {
    const int nrow = n + 1;
    constexpr int B = gp_block<NP>::B;
    const int lane = W::lane();
    constexpr bool PER_WAVE = (W::NWAVES <= 4);             // else: wave 0 inverts one shared copy
    double (*Pw)[B] = S.P[PER_WAVE ? W::wave_id() : 0];
    double ld = 0.0;
    for (int k0 = 0; k0 < n; k0 += B) {
        const int bs = (n - k0 < B) ? n - k0 : B;
        GP_T0();
        // (1) V[p][i] = A(i, k0+p) for every i (symmetric access); rows p >= bs are zero padding
        for (int idx = lane; idx < B * NP; idx += W::LANES) {
            const int p = idx / NP, i = idx - p * NP;       // NP is a compile-time constant
            if (i < nrow) {
                const int kp = k0 + p;
                S.V[p][i] = (p < bs) ? ((i >= kp) ? A[tri_index(i, kp)] : A[tri_index(kp, i)]) : 0.0;
            }
        }
        if (!PER_WAVE && lane == 0) S.pivot_bad = 0;
        W::sync();
        GP_T(0);
        // (2) inversion of the (identity-padded) pivot block by B single-index sweeps with only
        //     wave-level hand-offs: every wavefront on its own copy, or wave 0 on a shared one
        bool bad = false;
        if (PER_WAVE || W::wave_id() == 0) {
            bad = gp_pivot_inverse<W, NP>(S, Pw, k0, bs, ld);
            if (!PER_WAVE && bad && W::wlane() == 0) S.pivot_bad = 1;
        }
        if (!PER_WAVE) {
            W::sync();
            bad = (S.pivot_bad != 0);
            // log-determinant: only wave 0 accumulated it; every lane needs the same value
            ld = W::bcast_from_first_wave(ld);
        }
        if (bad) return false;
        GP_T(1);
        // now Pw = -A_PP^-1 (padding: -1 on the diagonal, met only by zero rows of V)
        // (3) Wm[p][i] = sum_q V[q][i] * Pinv[q][p]   (Pinv = -Pw)
        gp_row_weights<W, NP>(S, Pw, nrow);
        W::sync();
        GP_T(2);
        // (4) rank-B update of every tile: C -= Wm(rows of the tile) * V(columns of the tile)^T.
        //     Elements in pivot rows / columns are not stored (they are rewritten in (5)).
        gp_tile_update<W, NP, KP>(A, nrow, S, k0, bs);
        GP_T(8);
        // (5) new pivot rows / columns:  A_RP <- A_RP A_PP^-1 ,  A_PP <- -A_PP^-1  (disjoint from (4))
        for (int idx = lane; idx < B * NP; idx += W::LANES) {
            const int p = idx / NP, i = idx - p * NP;
            if (i < nrow && p < bs) {
                const int kp = k0 + p;
                const bool in_blk = (i >= k0 && i < k0 + bs);
                const double v = in_blk ? Pw[i - k0][p] : S.Wm[p][i];
                if (i >= kp) A[tri_index(i, kp)] = v; else if (!in_blk) A[tri_index(kp, i)] = v;
            }
        }
        W::sync();
        GP_T(3);
    }
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

// One evaluation of f = -log-likelihood and its gradient at p (george GP.log_likelihood /
// grad_log_likelihood as wrapped by multiband_gp.py:141-154).  K is overwritten (by -K^-1).  On a
// failed factorisation f = 1e25 and g = 0.  `need_grad` false: only alpha and f (prediction pass).
template <class W, int NP, class KP>
LCFE_FN_NOINLINE void gp_eval(const double* p, int n, GpLds<NP, W::NWAVES>& S, KP K, double& f, double* g, bool need_grad) {
    const int lane = W::lane();
    constexpr int G = (W::LANES >= 64) ? 64 : W::LANES;
    constexpr int RG = W::LANES / G;
    const int rl = lane / G, cl = lane % G;
    const double mu = p[0], c = exp(p[1]), m0 = exp(p[2]), m1 = exp(p[3]);
    GP_T0();
    // Gram matrix, packed lower
    for (int i = rl; i < n; i += RG) {
        const double ti = S.t[i], li = S.lam[i];
        for (int j = cl; j <= i; j += G) {
            const double dt = ti - S.t[j], dl = li - S.lam[j];
            double e;
            double k = gp_kernel(dt * dt, dl * dl, c, m0, m1, e);
            if (j == i) k += S.e2[i] + GP_TINY;
            K[tri_index(i, j)] = k;
        }
    }
    for (int i = lane; i <= n; i += W::LANES) K[tri_index(n, i)] = (i < n) ? S.y[i] - mu : 0.0;   // augmented row
    W::sync();
    GP_T(4);
    double logdet;
    g[0] = g[1] = g[2] = g[3] = 0.0;
    if (!gp_sweep_inverse<W, NP, KP>(K, n, S, logdet)) { f = 1e25; W::sync(); return; }
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
    for (int i = rl; i < n; i += RG) {
        const double ai = S.alpha[i], ti = S.t[i], li = S.lam[i];
        for (int j = cl; j <= i; j += G) {
            const double dt = ti - S.t[j], dl = li - S.lam[j];
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
template <class W, int NP, class Ev>
LCFE_FN void gp_object(const ObjIn& L, GpLds<NP, W::NWAVES>& S, Ev&& gp_ev, int32_t* st) {
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
            S.lam[pos] = gp_wavelength(L.b[i]);
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
    // values occupy ranks 0..nnz-1 (r[] is scratch here)
    for (int i = lane; i < n; i += W::LANES) {
        const bool nz = (S.y[i] != 0.0);
        S.r[i] = nz ? fabs(S.y[i]) : __builtin_inf();
        nnz += nz ? 1 : 0;
    }
    nnz = W::sum(nnz);
    W::sync();
    double scale = qnan();
    if (nnz > 0) {
        wave_rank_select<W>(S.r, n, (nnz - 1) / 2, nnz / 2, S.slot, reinterpret_cast<unsigned long long*>(S.alpha));
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
    int n_iter = 0, n_eval = 0, why = LB_ERROR;
    if (finite0) {
        auto ev = [&](const double* x, double& f, double* g) { gp_ev(x, n, f, g, true); };
        why = lbfgsb_minimize<4, 10>(p, fval, ev, 100, 1e7, 1e-5, 20, n_iter, n_eval, S.lb_s, S.lb_y, S.lb_rho);
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
    // ---- interpolate_multiband (:196-289): alpha at the optimum, then 12 predictions
    double ftmp, gtmp[4];
    gp_ev(p, n, ftmp, gtmp, false);
    if (ftmp >= 1e25) { W::sync(); return; }          // factorisation failed at the optimum: predict raises -> NaN (:279-287)
    const double c = exp(p[1]), m0 = exp(p[2]), m1 = exp(p[3]);
    const double EP[4] = {0, 20, 50, 100};
    const int PB[3] = {1, 2, 3};
    double fl[12];
    for (int q = 0; q < 12; ++q) {
        const double tp = peak_time + EP[q / 3], lp = gp_wavelength(PB[q % 3]);
        double s = 0;
        for (int i = lane; i < n; i += W::LANES) {
            const double dt = tp - S.t[i], dl = lp - S.lam[i];
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
