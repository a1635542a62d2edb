// lbfgsb_box.hpp -- L-BFGS-B WITH box bounds for a handful of variables, as scikit-learn's
// GaussianProcessRegressor reaches it through ``scipy.optimize.minimize(method='L-BFGS-B',
// jac=True, bounds=...)`` (sklearn/gaussian_process/_gpr.py::_constrained_optimization; scipy
// defaults maxcor=10, ftol=2.22e-9 -> factr=1e7, gtol(pgtol)=1e-5, maxfun=maxiter=15000, maxls=20).
//
// Groundwork for the per-band GP of src/features/gaussian_process.py: not
// yet used by a kernel; tests/test_lbfgsb_box.py checks it against scipy itself on the host build.
//
// Restated from L-BFGS-B 3.0 (Zhu, Byrd, Lu, Nocedal; Morales & Nocedal 2011), subroutines
// active / projgr / cauchy / freev / cmprlb / subsm / lnsrlb / matupd of lbfgsb.f, with ONE
// deliberate difference: the limited-memory matrix B = theta I - W M W' is formed as a dense N x N
// matrix (N <= 4 here) by applying the stored BFGS pairs to theta I, instead of through the
// compact 2m x 2m middle matrix.  The two are the same matrix in exact arithmetic
// (Byrd, Nocedal, Schnabel 1994), so the iterates agree with scipy's to rounding.
//   * generalised Cauchy point: piecewise-linear projected-gradient path, one breakpoint per
//     bounded variable, f' and f'' of the quadratic model re-evaluated on every segment
//   * subspace minimisation over the variables free at the Cauchy point: Newton step of the
//     model, projected onto the box; if the projected point is not a descent direction, the
//     classical truncated step (3.0's "projection, then backtrack" rule)
//   * line search dcsrch (lbfgsb.hpp) with stpmx = largest feasible step, first trial step 1
//     (all variables are boxed), x := z when the step is exactly 1
//   * BFGS pair skipped when s'y <= eps * (-g_old's); memory dropped when the line search fails
//   * stop on the projected gradient, on (f_old - f) <= factr*eps*max(|f_old|,|f|,1), on maxiter
//     (scipy's driver, tested at every new iterate) or maxfun.
// Block-uniform scalar code like lbfgsb.hpp: `eval(x, f, g)` is collective.
#pragma once
#include "lbfgsb.hpp"

namespace lcfe {

enum { LBB_MAXFUN = 6 };

template <int N>
LCFE_FN double lbb_projgr(const double x[N], const double g[N], const double lo[N], const double hi[N]) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double gi = g[i];
        if (gi < 0.0) gi = fmax(x[i] - hi[i], gi); else gi = fmin(x[i] - lo[i], gi);
        s = fmax(s, fabs(gi));
    }
    return s;
}

// B = theta I updated with the `col` stored pairs, oldest first
template <int N, int M>
LCFE_FN void lbb_dense_b(double B[N][N], double theta, int col, int head, double (*Sm)[N], double (*Ym)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) B[i][j] = (i == j) ? theta : 0.0;
    for (int k = 0; k < col; ++k) {
        const int idx = (head + k) % M;
        double Bs[N], sBs = 0, sy = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double a = 0;
#pragma unroll
            for (int j = 0; j < N; ++j) a += B[i][j] * Sm[idx][j];
            Bs[i] = a;
        }
#pragma unroll
        for (int i = 0; i < N; ++i) { sBs += Sm[idx][i] * Bs[i]; sy += Sm[idx][i] * Ym[idx][i]; }
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) B[i][j] += -Bs[i] * Bs[j] / sBs + Ym[idx][i] * Ym[idx][j] / sy;
    }
}

// lbfgsb.f cauchy: generalised Cauchy point z of the quadratic model along the projected
// steepest-descent path from x; where[i] = 1 / 2 (at lower / upper bound), 0 (free), -3 (free, g = 0)
template <int N>
LCFE_FN void lbb_cauchy(const double x[N], const double g[N], const double lo[N], const double hi[N],
                        const double B[N][N], double sbgnrm, double z[N], int where[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) z[i] = x[i];
    if (sbgnrm <= 0.0) return;
    double d[N], tb[N];
    bool has_bp[N];
    int nbreak = 0;
    bool bnded = true, any_dir = false;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double neggi = -g[i];
        const double tl = x[i] - lo[i], tu = hi[i] - x[i];
        const bool xlower = tl <= 0.0, xupper = tu <= 0.0;
        where[i] = 0;
        if (xlower) { if (neggi <= 0.0) where[i] = 1; }
        else if (xupper) { if (neggi >= 0.0) where[i] = 2; }
        else if (fabs(neggi) <= 0.0) where[i] = -3;
        has_bp[i] = false;
        tb[i] = 0.0;
        if (where[i] != 0) { d[i] = 0.0; continue; }
        d[i] = neggi;
        any_dir = true;
        if (neggi < 0.0) { has_bp[i] = true; tb[i] = tl / (-neggi); ++nbreak; }
        else if (neggi > 0.0) { has_bp[i] = true; tb[i] = tu / neggi; ++nbreak; }
        else if (fabs(neggi) > 0.0) bnded = false;
    }
    (void)bnded;
    if (!any_dir) return;
    // model derivatives along d from the current point x + zz:  f1 = g'd + d'B zz,  f2 = d'B d
    double zz[N];
#pragma unroll
    for (int i = 0; i < N; ++i) zz[i] = 0.0;
    auto derivs = [&](double& f1, double& f2) {
        f1 = 0; f2 = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double bd = 0, bz = 0;
#pragma unroll
            for (int j = 0; j < N; ++j) { bd += B[i][j] * d[j]; bz += B[i][j] * zz[j]; }
            f1 += d[i] * (g[i] + bz);
            f2 += d[i] * bd;
        }
    };
    double f1, f2;
    derivs(f1, f2);
    const double f2_org = f2;
    double dtm = -f1 / f2, tsum = 0.0, tj = 0.0;
    int nleft = nbreak;
    bool all_fixed = false;
    while (nleft > 0) {
        // next smallest breakpoint among the variables still moving
        int ibp = -1;
        double tmin = 0;
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (has_bp[i] && (ibp < 0 || tb[i] < tmin)) { ibp = i; tmin = tb[i]; }
        const double dt = tmin - tj;
        if (dtm < dt) break;
        tj = tmin;
        tsum += dt;
        --nleft;
        has_bp[ibp] = false;
        const double dibp = d[ibp];
        // the other moving variables advance by dt along d; this one lands on its bound
#pragma unroll
        for (int i = 0; i < N; ++i) zz[i] += dt * d[i];
        d[ibp] = 0.0;
        if (dibp > 0.0) { z[ibp] = hi[ibp]; zz[ibp] = hi[ibp] - x[ibp]; where[ibp] = 2; }
        else { z[ibp] = lo[ibp]; zz[ibp] = lo[ibp] - x[ibp]; where[ibp] = 1; }
        if (nleft == 0 && nbreak == N) { all_fixed = true; break; }
        derivs(f1, f2);
        f2 = fmax(LB_EPS * f2_org, f2);
        dtm = -f1 / f2;
    }
    if (all_fixed) return;
    if (dtm <= 0.0) dtm = 0.0;
    // free variables and variables whose breakpoint was not reached: x + (tsum + dtm) d
    const double tt = tsum + dtm;
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (d[i] != 0.0) z[i] = x[i] + tt * d[i];
}

// lbfgsb.f cmprlb + subsm: Newton step of the model over the variables free at the Cauchy point z,
// projected onto the box (3.0); z is updated in place.
template <int N>
LCFE_FN void lbb_subspace(const double x[N], const double g[N], const double lo[N], const double hi[N],
                          const double B[N][N], const int where[N], double z[N]) {
    int ind[N], nsub = 0;
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (where[i] <= 0) ind[nsub++] = i;
    if (nsub == 0) return;
    // r = -(g + B (z - x)) on the free set
    double r[N], A[N][N], dd[N];
    for (int a = 0; a < nsub; ++a) {
        const int i = ind[a];
        double bz = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) bz += B[i][j] * (z[j] - x[j]);
        r[a] = -(g[i] + bz);
        for (int b = 0; b < nsub; ++b) A[a][b] = B[i][ind[b]];
    }
    // solve A dd = r (A symmetric positive definite, nsub <= N): Gaussian elimination
    for (int k = 0; k < nsub; ++k) {
        for (int a = k + 1; a < nsub; ++a) {
            const double m = A[a][k] / A[k][k];
            for (int b = k; b < nsub; ++b) A[a][b] -= m * A[k][b];
            r[a] -= m * r[k];
        }
    }
    for (int a = nsub - 1; a >= 0; --a) {
        double s = r[a];
        for (int b = a + 1; b < nsub; ++b) s -= A[a][b] * dd[b];
        dd[a] = s / A[a][a];
    }
    // projection of z + dd onto the box
    double zp[N];
#pragma unroll
    for (int i = 0; i < N; ++i) zp[i] = z[i];
    bool projected = false;
    for (int a = 0; a < nsub; ++a) {
        const int k = ind[a];
        double v = fmax(lo[k], z[k] + dd[a]);
        v = fmin(hi[k], v);
        z[k] = v;
        if (v == lo[k] || v == hi[k]) projected = true;
    }
    if (!projected) return;
    double dd_p = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) dd_p += (z[i] - x[i]) * g[i];
    if (dd_p <= 0.0) return;
    // not a descent direction: classical truncated step from the Cauchy point
#pragma unroll
    for (int i = 0; i < N; ++i) z[i] = zp[i];
    double alpha = 1.0, temp1 = 1.0;
    int ibd = -1;
    for (int a = 0; a < nsub; ++a) {
        const int k = ind[a];
        const double dk = dd[a];
        if (dk < 0.0) {
            const double t2 = lo[k] - z[k];
            if (t2 >= 0.0) temp1 = 0.0; else if (dk * alpha < t2) temp1 = t2 / dk;
        } else if (dk > 0.0) {
            const double t2 = hi[k] - z[k];
            if (t2 <= 0.0) temp1 = 0.0; else if (dk * alpha > t2) temp1 = t2 / dk;
        }
        if (temp1 < alpha) { alpha = temp1; ibd = a; }
    }
    if (alpha < 1.0 && ibd >= 0) {
        const int k = ind[ibd];
        if (dd[ibd] > 0.0) { z[k] = hi[k]; dd[ibd] = 0.0; }
        else if (dd[ibd] < 0.0) { z[k] = lo[k]; dd[ibd] = 0.0; }
    }
    for (int a = 0; a < nsub; ++a) z[ind[a]] += alpha * dd[a];
}

// Minimise f over the box [lo, hi].  x is projected onto the box first (lbfgsb.f `active`).
// Sm, Ym: M rows of caller-provided (block-shared) storage for the BFGS pairs.
template <int N, int M, class Eval>
LCFE_FN int lbfgsb_box_minimize(double x[N], const double lo[N], const double hi[N], double& f, Eval&& eval,
                                int maxiter, int maxfun, double factr, double pgtol, int maxls, int& n_iter,
                                int& n_eval, double (*Sm)[N], double (*Ym)[N]) {
    double g[N], d[N], t[N], r[N], z[N], B[N][N];
    int where[N];
    int col = 0, head = 0;
    double theta = 1.0;
    n_iter = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = fmin(hi[i], fmax(lo[i], x[i]));
    eval(x, f, g);
    n_eval = 1;
    double sbgnrm = lbb_projgr<N>(x, g, lo, hi);
    if (sbgnrm <= pgtol) return LB_CONVERGED_PG;
    while (true) {
        lbb_dense_b<N, M>(B, theta, col, head, Sm, Ym);
        lbb_cauchy<N>(x, g, lo, hi, B, sbgnrm, z, where);
        if (col > 0) lbb_subspace<N>(x, g, lo, hi, B, where, z);
#pragma unroll
        for (int i = 0; i < N; ++i) d[i] = z[i] - x[i];
        // ---- lnsrlb
        double dtd = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) dtd += d[i] * d[i];
        double stpmx = 1e10;
        if (n_iter == 0) stpmx = 1.0;
        else {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double a1 = d[i];
                if (a1 < 0.0) {
                    const double a2 = lo[i] - x[i];
                    if (a2 >= 0.0) stpmx = 0.0; else if (a1 * stpmx < a2) stpmx = a2 / a1;
                } else if (a1 > 0.0) {
                    const double a2 = hi[i] - x[i];
                    if (a2 <= 0.0) stpmx = 0.0; else if (a1 * stpmx > a2) stpmx = a2 / a1;
                }
            }
        }
        double stp = 1.0;                                       // every variable is boxed
#pragma unroll
        for (int i = 0; i < N; ++i) { t[i] = x[i]; r[i] = g[i]; }
        const double fold = f;
        int ifun = 0, iback = 0;
        double gd = 0, gdold = 0;
        Dcsrch ls;
        bool ls_fail = false, first = true, out_of_evals = false;
        while (true) {
            gd = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) gd += g[i] * d[i];
            if (ifun == 0) {
                gdold = gd;
                if (gd >= 0.0) { ls_fail = true; break; }
            }
            const int task = dcsrch(f, gd, stp, 1e-3, 0.9, 0.1, 0.0, stpmx, first, ls);
            first = false;
            if (task == LS_ERROR) { ls_fail = true; break; }
            if (task == LS_CONV || task == LS_WARN) break;
            ++ifun;
            iback = ifun - 1;
            if (iback >= maxls) { ls_fail = true; break; }
            if (stp == 1.0) {
#pragma unroll
                for (int i = 0; i < N; ++i) x[i] = z[i];
            } else {
#pragma unroll
                for (int i = 0; i < N; ++i) x[i] = stp * d[i] + t[i];
            }
            eval(x, f, g);
            ++n_eval;
            if (n_eval > maxfun) { out_of_evals = true; }
        }
        if (ls_fail) {
#pragma unroll
            for (int i = 0; i < N; ++i) { x[i] = t[i]; g[i] = r[i]; }
            f = fold;
            if (col == 0) return LB_ABNORMAL;
            col = 0; head = 0; theta = 1.0;
            continue;
        }
        ++n_iter;
        sbgnrm = lbb_projgr<N>(x, g, lo, hi);
        if (sbgnrm <= pgtol) return LB_CONVERGED_PG;
        const double ddum0 = max3(fabs(fold), fabs(f), 1.0);
        if ((fold - f) <= LB_EPS * factr * ddum0) return LB_CONVERGED_F;
        if (n_iter >= maxiter) return LB_MAXITER;
        if (out_of_evals) return LBB_MAXFUN;
        // ---- BFGS pair
        double rr = 0, dr, ddum;
#pragma unroll
        for (int i = 0; i < N; ++i) { r[i] = g[i] - r[i]; rr += r[i] * r[i]; }
        if (stp == 1.0) { dr = gd - gdold; ddum = -gdold; }
        else {
            dr = (gd - gdold) * stp;
#pragma unroll
            for (int i = 0; i < N; ++i) d[i] *= stp;
            ddum = -gdold * stp;
        }
        if (dr <= LB_EPS * ddum) continue;
        int slot;
        if (col < M) { slot = (head + col) % M; ++col; }
        else { slot = head; head = (head + 1) % M; }
#pragma unroll
        for (int i = 0; i < N; ++i) { Sm[slot][i] = d[i]; Ym[slot][i] = r[i]; }
        theta = rr / dr;
    }
}

}  // namespace lcfe
