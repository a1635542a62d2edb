// gp_reg.hpp -- register-resident variant of the GP evaluation (see gp.hpp for the algorithm).
//
// The packed Gram matrix of one light curve is held in the REGISTERS of the workgroup instead of
// LDS/global memory: the W::LANES = TS x TS threads form a square grid, thread (r, c) owns the
// elements A(ta*TS + r, tb*TS + c) of every tile (ta, tb), tb <= ta, of a T x T tiling (a 2-D
// block-cyclic layout, T(T+1)/2 doubles per thread).  The blocked symmetric sweep then becomes a
// register-tiled outer product: per pivot block the 8 pivot columns pass through LDS (V, Wm) and
// every thread applies  a[ta][tb] -= Wm[p][row] * V[p][col]  to its own registers -- no matrix
// traffic at all, 2T LDS reads per T(T+1)/2 FMAs.
//
// The matrix is augmented by one row holding the residual r = y - mu (never pivoted): after the
// sweep that row holds alpha = K^-1 r and its diagonal entry -r'K^-1 r (the classic regression use
// of the sweep operator), so no triangular solves or matrix-vector products are needed.
#pragma once
#include "gp.hpp"

namespace lcfe {

template <int T>
LCFE_HD constexpr int gp_slot(int ta, int tb) { return ta * (ta + 1) / 2 + tb; }

// f = -log-likelihood and gradient at p; same contract as gp_eval.  NP = LDS array capacity
// (>= n + 1), TS*TS = W::LANES, T*TS >= n + 1.
template <class W, int NP, int TS, int T>
LCFE_FN_NOINLINE void gp_eval_reg(const double* p, int n, GpLds<NP, W::NWAVES>& S, double& f, double* g, bool need_grad) {
    constexpr int B = gp_block<NP>::B;
    static_assert(W::LANES == TS * TS, "square thread grid");
    constexpr int NSLOT = T * (T + 1) / 2;
    const int lane = W::lane();
    const int r = lane / TS, c = lane % TS;
    const double mu = p[0], cc = exp(p[1]), m0 = exp(p[2]), m1 = exp(p[3]);
    double a[NSLOT];
    // ---- Gram matrix (+ augmented residual row n) straight into registers
    for (int i = lane; i < n; i += W::LANES) S.r[i] = S.y[i] - mu;
    W::sync();
#pragma unroll
    for (int ta = 0; ta < T; ++ta) {
        const int i = ta * TS + r;
#pragma unroll
        for (int tb = 0; tb <= ta; ++tb) {
            const int j = tb * TS + c;
            double v = 0.0;
            if (j <= i && i < n) {
                const double dt = S.t[i] - S.t[j], dl = S.lam[i] - S.lam[j];
                double e;
                v = gp_kernel(dt * dt, dl * dl, cc, m0, m1, e);
                if (j == i) v += S.e2[i] + GP_TINY;
            } else if (i == n && j < n) {
                v = S.r[j];
            }
            a[gp_slot<T>(ta, tb)] = v;
        }
    }
    g[0] = g[1] = g[2] = g[3] = 0.0;
    double (*Pw)[B] = S.P[W::wave_id()];
    double ld = 0.0;
    bool ok = true;
    const int nrow = n + 1;                                   // rows incl. the augmented one
    for (int k0 = 0; k0 < n; k0 += B) {
        const int bs = (n - k0 < B) ? n - k0 : B;
        // (1) pivot columns -> LDS:  V[p][i] = A(i, k0+p)  (column part i >= kp from the owner of
        //     column kp, row part i < kp from the owner of row kp)
        {
#pragma unroll
            for (int ta = 0; ta < T; ++ta) {
                const int i = ta * TS + r;
                const bool i_in = (i >= k0 && i < k0 + bs);
#pragma unroll
                for (int tb = 0; tb <= ta; ++tb) {
                    const int j = tb * TS + c;
                    const bool j_in = (j >= k0 && j < k0 + bs);
                    const double v = a[gp_slot<T>(ta, tb)];
                    if (j_in && i >= j && i < nrow) S.V[j - k0][i] = v;
                    if (i_in && j < i) S.V[i - k0][j] = v;
                }
            }
            // zero padding rows of a short last block
            for (int idx = lane; idx < (B - bs) * NP; idx += W::LANES) S.V[bs + idx / NP][idx % NP] = 0.0;
        }
        W::sync();
        // (2) per-wave inversion of the identity-padded pivot block (only wave-level hand-offs)
        for (int e = W::wlane(); e < B * B; e += W::WAVE) {
            const int x = e / B, y = e % B;
            Pw[x][y] = (x < bs && y < bs) ? S.V[y][k0 + x] : ((x == y) ? 1.0 : 0.0);
        }
        W::wave_sync();
        double prod = 1.0;
        for (int q = 0; q < B; ++q) {
            const double d = Pw[q][q];
            if (!(d > 0.0)) ok = false;                       // identical in every wavefront
            prod *= d;
            if ((q & 7) == 7) { ld += log(prod); prod = 1.0; }
            const double inv = 1.0 / d;
            constexpr int NV = (B * B + W::WAVE - 1) / W::WAVE;
            double nv[NV];
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int e = W::wlane() + u * W::WAVE;
                const int x = e / B, y = e % B;
                const double pxq = Pw[x][q], pqy = Pw[q][y], pxy = Pw[x][y];
                double v = pxy - pxq * pqy * inv;
                if (x == q || y == q) v = ((x == q) ? pqy : pxq) * inv;
                if (x == q && y == q) v = -inv;
                nv[u] = v;
            }
            W::wave_sync();
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int e = W::wlane() + u * W::WAVE;
                Pw[e / B][e % B] = nv[u];
            }
            W::wave_sync();
        }
        if (!ok) break;                                       // uniform
        // (3) Wm[p][i] = sum_q V[q][i] * Pinv[q][p]   (Pinv = -Pw)
        for (int i = lane; i < nrow; i += W::LANES) {
            double vq[B];
#pragma unroll
            for (int q = 0; q < B; ++q) vq[q] = S.V[q][i];
#pragma unroll
            for (int pp = 0; pp < B; ++pp) {
                double sacc = 0;
#pragma unroll
                for (int q = 0; q < B; ++q) sacc = fma(-vq[q], Pw[q][pp], sacc);
                S.Wm[pp][i] = sacc;
            }
        }
        W::sync();
        // (4) register-tiled rank-B update, one pivot at a time: a[ta][tb] -= Wm[p][row] * V[p][col]
#pragma unroll
        for (int pp = 0; pp < B; ++pp) {
            double wr[T], vc[T];
#pragma unroll
            for (int ta = 0; ta < T; ++ta) {
                const int i = ta * TS + r, j = ta * TS + c;
                wr[ta] = (i < nrow) ? S.Wm[pp][i] : 0.0;
                vc[ta] = (j < nrow) ? S.V[pp][j] : 0.0;
            }
#pragma unroll
            for (int ta = 0; ta < T; ++ta)
#pragma unroll
                for (int tb = 0; tb <= ta; ++tb) a[gp_slot<T>(ta, tb)] = fma(-wr[ta], vc[tb], a[gp_slot<T>(ta, tb)]);
        }
        // (5) new pivot rows / columns:  A_RP <- A_RP A_PP^-1 ,  A_PP <- -A_PP^-1
        {
#pragma unroll
            for (int ta = 0; ta < T; ++ta) {
                const int i = ta * TS + r;
                const bool i_in = (i >= k0 && i < k0 + bs);
#pragma unroll
                for (int tb = 0; tb <= ta; ++tb) {
                    const int j = tb * TS + c;
                    if (j > i || i >= nrow) continue;
                    const bool j_in = (j >= k0 && j < k0 + bs);
                    if (j_in && i_in) a[gp_slot<T>(ta, tb)] = Pw[i - k0][j - k0];
                    else if (j_in) a[gp_slot<T>(ta, tb)] = S.Wm[j - k0][i];
                    else if (i_in) a[gp_slot<T>(ta, tb)] = S.Wm[i - k0][j];
                }
            }
        }
        W::sync();
    }
    if (!ok) { f = 1e25; W::sync(); return; }
    // ---- alpha = augmented row, r'K^-1 r = -A(n, n)
    {
        const int tan = n / TS, rn = n % TS;
#pragma unroll
        for (int ta = 0; ta < T; ++ta)
#pragma unroll
            for (int tb = 0; tb <= ta; ++tb) {
                const int j = tb * TS + c;
                if (ta == tan && r == rn) {
                    if (j < n) S.alpha[j] = a[gp_slot<T>(ta, tb)];
                    else if (j == n) S.slot[0] = -a[gp_slot<T>(ta, tb)];
                }
            }
    }
    W::sync();
    const double ra = S.slot[0];
    double sa = 0;
    for (int i = lane; i < n; i += W::LANES) sa += S.alpha[i];
    sa = W::sum(sa);
    const double ll = -0.5 * (ra + ld + n * GP_LOG_2PI);
    f = finite_d(ll) ? -ll : 1e25;
    if (!need_grad) { W::sync(); return; }
    // ---- gradient: 0.5 * sum_ij (alpha_i alpha_j - Kinv_ij) dK_ij/dtheta , Kinv_ij = -a
    double g1 = 0, g2 = 0, g3 = 0;
#pragma unroll
    for (int ta = 0; ta < T; ++ta) {
        const int i = ta * TS + r;
#pragma unroll
        for (int tb = 0; tb <= ta; ++tb) {
            const int j = tb * TS + c;
            if (j <= i && i < n) {
                const double dt = S.t[i] - S.t[j], dl = S.lam[i] - S.lam[j];
                const double dt2 = dt * dt, dl2 = dl * dl;
                double e;
                const double k = gp_kernel(dt2, dl2, cc, m0, m1, e);
                const double w = (S.alpha[i] * S.alpha[j] + a[gp_slot<T>(ta, tb)]) * ((j == i) ? 1.0 : 2.0);
                g1 += w * k;
                g2 += w * e * dt2 / m0;
                g3 += w * e * dl2 / m1;
            }
        }
    }
    g1 = W::sum(g1);
    g2 = W::sum(g2);
    g3 = W::sum(g3);
    g[0] = -sa;
    g[1] = -0.5 * g1;
    g[2] = -0.5 * g2;
    g[3] = -0.5 * g3;
    W::sync();
}

}  // namespace lcfe
