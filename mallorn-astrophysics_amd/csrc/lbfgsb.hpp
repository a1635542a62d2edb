// lbfgsb.hpp -- unbounded L-BFGS-B as the reference reaches it through
// ``scipy.optimize.minimize(method='L-BFGS-B', jac=..., options={'maxiter': 100})``
// (multiband_gp.py:158-164; defaults maxcor=10, ftol=2.22e-9 -> factr=1e7, gtol(pgtol)=1e-5,
// maxls=20: scipy/optimize/_lbfgsb_py.py:290-294).
//
// Restated from L-BFGS-B 3.0 (Zhu, Byrd, Lu, Nocedal; Morales & Nocedal 2011) for the case the
// reference uses -- no bounds:
//   * first iteration (and after a memory reset): d = -g, first trial step 1/||d||   (cauchy with an
//     empty breakpoint set; lnsrlb "iter == 0 && !boxed")
//   * later iterations: d = -H g with the limited-memory BFGS inverse Hessian, H0 = I/theta,
//     theta = y'y / s'y  (subsm over the whole space == two-loop recursion), trial step 1
//   * line search dcsrch/dcstep (More'-Thuente, MINPACK-2) with ftol=1e-3, gtol=0.9, xtol=0.1,
//     stpmin=0, stpmax=1e10; at most maxls=20 evaluations per iteration, else the memory is reset
//     (or the run ends abnormally if it was already empty)
//   * update skipped when s'y <= eps * (-g_old's)
//   * stop on max|g_i| <= pgtol, on (f_old - f) <= factr*eps*max(|f_old|,|f|,1), or after
//     `maxiter` iterations (scipy's driver loop).
// The state machine is reverse-communication like the original: the caller evaluates f and g.
// Everything here is wave/block-uniform scalar code (n = 4 parameters).
#pragma once
#include <type_traits>
#include "wave.hpp"

namespace lcfe {

constexpr double LB_EPS = 2.220446049250313e-16;

struct Dcsrch {
    // saved state of MINPACK-2 dcsrch
    bool brackt;
    int stage;
    double ginit, gtest, gx, gy, finit, fx, fy, stx, sty, stmin, stmax, width, width1;
};

enum { LS_FG = 0, LS_CONV = 1, LS_WARN = 2, LS_ERROR = 3 };

LCFE_FN double max3(double a, double b, double c) { return fmax(fmax(a, b), c); }

// MINPACK-2 dcstep
LCFE_FN void dcstep(double& stx, double& fx, double& dx, double& sty, double& fy, double& dy, double& stp,
                    double fp, double dp, bool& brackt, double stpmin, double stpmax) {
    const double sgnd = dp * (dx / fabs(dx));
    double stpf;
    if (fp > fx) {
        const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
        const double s = max3(fabs(theta), fabs(dx), fabs(dp));
        double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
        if (stp < stx) gamma = -gamma;
        const double p = (gamma - dx) + theta;
        const double q = ((gamma - dx) + gamma) + dp;
        const double r = p / q;
        const double stpc = stx + r * (stp - stx);
        const double stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
        if (fabs(stpc - stx) < fabs(stpq - stx)) stpf = stpc;
        else stpf = stpc + (stpq - stpc) / 2.0;
        brackt = true;
    } else if (sgnd < 0.0) {
        const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
        const double s = max3(fabs(theta), fabs(dx), fabs(dp));
        double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
        if (stp > stx) gamma = -gamma;
        const double p = (gamma - dp) + theta;
        const double q = ((gamma - dp) + gamma) + dx;
        const double r = p / q;
        const double stpc = stp + r * (stx - stp);
        const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
        if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc;
        else stpf = stpq;
        brackt = true;
    } else if (fabs(dp) < fabs(dx)) {
        const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
        const double s = max3(fabs(theta), fabs(dx), fabs(dp));
        double gamma = s * sqrt(fmax(0.0, (theta / s) * (theta / s) - (dx / s) * (dp / s)));
        if (stp > stx) gamma = -gamma;
        const double p = (gamma - dp) + theta;
        const double q = (gamma + (dx - dp)) + gamma;
        const double r = p / q;
        double stpc;
        if (r < 0.0 && gamma != 0.0) stpc = stp + r * (stx - stp);
        else if (stp > stx) stpc = stpmax;
        else stpc = stpmin;
        const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
        if (brackt) {
            if (fabs(stpc - stp) < fabs(stpq - stp)) stpf = stpc;
            else stpf = stpq;
            if (stp > stx) stpf = fmin(stp + 0.66 * (sty - stp), stpf);
            else stpf = fmax(stp + 0.66 * (sty - stp), stpf);
        } else {
            if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc;
            else stpf = stpq;
            stpf = fmin(stpmax, stpf);
            stpf = fmax(stpmin, stpf);
        }
    } else {
        if (brackt) {
            const double theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
            const double s = max3(fabs(theta), fabs(dy), fabs(dp));
            double gamma = s * sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
            if (stp > sty) gamma = -gamma;
            const double p = (gamma - dp) + theta;
            const double q = ((gamma - dp) + gamma) + dy;
            const double r = p / q;
            stpf = stp + r * (sty - stp);
        } else if (stp > stx) stpf = stpmax;
        else stpf = stpmin;
    }
    if (fp > fx) {
        sty = stp; fy = fp; dy = dp;
    } else {
        if (sgnd < 0.0) { sty = stx; fy = fx; dy = dx; }
        stx = stp; fx = fp; dx = dp;
    }
    stp = stpf;
}

// MINPACK-2 dcsrch.  start == true: first call of a line search.
LCFE_FN int dcsrch(double f, double g, double& stp, double ftol, double gtol, double xtol, double stpmin,
                   double stpmax, bool start, Dcsrch& S) {
    const double xtrapl = 1.1, xtrapu = 4.0;
    if (start) {
        if (stp < stpmin || stp > stpmax || g >= 0.0) return LS_ERROR;
        S.brackt = false;
        S.stage = 1;
        S.finit = f;
        S.ginit = g;
        S.gtest = ftol * S.ginit;
        S.width = stpmax - stpmin;
        S.width1 = S.width / 0.5;
        S.stx = 0.0; S.fx = S.finit; S.gx = S.ginit;
        S.sty = 0.0; S.fy = S.finit; S.gy = S.ginit;
        S.stmin = 0.0;
        S.stmax = stp + xtrapu * stp;
        return LS_FG;
    }
    const double ftest = S.finit + stp * S.gtest;
    if (S.stage == 1 && f <= ftest && g >= 0.0) S.stage = 2;
    int task = LS_FG;
    if (S.brackt && (stp <= S.stmin || stp >= S.stmax)) task = LS_WARN;
    if (S.brackt && S.stmax - S.stmin <= xtol * S.stmax) task = LS_WARN;
    if (stp == stpmax && f <= ftest && g <= S.gtest) task = LS_WARN;
    if (stp == stpmin && (f > ftest || g >= S.gtest)) task = LS_WARN;
    if (f <= ftest && fabs(g) <= gtol * (-S.ginit)) task = LS_CONV;
    if (task != LS_FG) return task;
    if (S.stage == 1 && f <= S.fx && f > ftest) {
        const double fm = f - stp * S.gtest;
        double fxm = S.fx - S.stx * S.gtest, fym = S.fy - S.sty * S.gtest;
        const double gm = g - S.gtest;
        double gxm = S.gx - S.gtest, gym = S.gy - S.gtest;
        dcstep(S.stx, fxm, gxm, S.sty, fym, gym, stp, fm, gm, S.brackt, S.stmin, S.stmax);
        S.fx = fxm + S.stx * S.gtest;
        S.fy = fym + S.sty * S.gtest;
        S.gx = gxm + S.gtest;
        S.gy = gym + S.gtest;
    } else {
        dcstep(S.stx, S.fx, S.gx, S.sty, S.fy, S.gy, stp, f, g, S.brackt, S.stmin, S.stmax);
    }
    if (S.brackt) {
        if (fabs(S.sty - S.stx) >= 0.66 * S.width1) stp = S.stx + 0.5 * (S.sty - S.stx);
        S.width1 = S.width;
        S.width = fabs(S.sty - S.stx);
    }
    if (S.brackt) {
        S.stmin = fmin(S.stx, S.sty);
        S.stmax = fmax(S.stx, S.sty);
    } else {
        S.stmin = stp + xtrapl * (stp - S.stx);
        S.stmax = stp + xtrapu * (stp - S.stx);
    }
    stp = fmax(stp, stpmin);
    stp = fmin(stp, stpmax);
    if ((S.brackt && (stp <= S.stmin || stp >= S.stmax)) || (S.brackt && S.stmax - S.stmin <= xtol * S.stmax))
        stp = S.stx;
    return LS_FG;
}

enum { LB_EVAL = 0, LB_CONVERGED_PG = 1, LB_CONVERGED_F = 2, LB_MAXITER = 3, LB_ABNORMAL = 4, LB_ERROR = 5 };

// Reverse-communication state of one minimisation over N variables with M correction pairs.  The
// caller evaluates f and g at `x`, stores them here and calls lb_advance(), which either asks for
// another evaluation (LB_EVAL, new trial point in `x`) or returns the stop reason (final iterate in
// `x`, `f`).  On the GPU the state lives in LDS and ONE thread advances it between two workgroup
// barriers: nothing of the optimiser is live in registers while the (wide, register-hungry)
// objective runs, so the objective's kernel is allocated for the linear algebra alone.
template <int N, int M>
struct LbState {
    double x[N], f, g[N];
    double d[N], t[N], r[N];
    double Sm[M][N], Ym[M][N], rho[M];      // circular memory of the (s, y) pairs
    double al[M];                           // two-loop recursion: the first loop's coefficients
    double theta, fold, stp, gdold;
    double factr, pgtol;
    Dcsrch ls;
    int col, head, n_iter, n_eval, ifun, phase, maxiter, maxls, first;
};

template <int N, int M>
LCFE_FN void lb_start(LbState<N, M>& S, const double* x0, int maxiter, double factr, double pgtol, int maxls) {
    for (int i = 0; i < N; ++i) S.x[i] = x0[i];
    S.col = 0; S.head = 0; S.theta = 1.0;
    S.n_iter = 0; S.n_eval = 0; S.phase = 0; S.ifun = 0; S.first = 1;
    S.maxiter = maxiter; S.maxls = maxls; S.factr = factr; S.pgtol = pgtol;
}

// member-wise copy of the line-search state between address spaces
template <class A, class B>
LCFE_FN void dcsrch_assign(A& a, const B& b) {
    a.brackt = b.brackt; a.stage = b.stage;
    a.ginit = b.ginit; a.gtest = b.gtest; a.gx = b.gx; a.gy = b.gy; a.finit = b.finit; a.fx = b.fx; a.fy = b.fy;
    a.stx = b.stx; a.sty = b.sty; a.stmin = b.stmin; a.stmax = b.stmax; a.width = b.width; a.width1 = b.width1;
}

// One step of the state machine after an evaluation of (f, g) at S.x.  Scalar code (one thread).
// IN_LDS (device): the caller's state is in LDS -- the function is not inlined, so without the address-space cast below
// every access to it is a flat load / store (130 of them, measured 10-20 k cycles per call in the 2-D GP kernels).
template <int N, int M, bool IN_LDS = false>
LCFE_FN_NOINLINE int lb_advance(LbState<N, M>& S0) {
#if defined(__HIP_DEVICE_COMPILE__)
    using StateRef = typename std::conditional<IN_LDS, __attribute__((address_space(3))) LbState<N, M>, LbState<N, M>>::type;
    StateRef& S = *(StateRef*)(&S0);
#else
    LbState<N, M>& S = S0;
#endif
    double x[N], g[N], d[N];
    double f = S.f;
#pragma unroll
    for (int i = 0; i < N; ++i) { x[i] = S.x[i]; g[i] = S.g[i]; d[i] = S.d[i]; }
    ++S.n_eval;
    const double stpmx = 1e10;
    bool in_search = (S.phase != 0);
    if (!in_search) {
        double sbgnrm = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) sbgnrm = fmax(sbgnrm, fabs(g[i]));
        if (sbgnrm <= S.pgtol) return LB_CONVERGED_PG;
    }
    while (true) {
        if (!in_search) {
            // ---- search direction d = -H g
            if (S.col == 0) {
#pragma unroll
                for (int i = 0; i < N; ++i) d[i] = -g[i] / S.theta;
            } else {
                double q[N];
#pragma unroll
                for (int i = 0; i < N; ++i) q[i] = g[i];
                for (int k = S.col - 1; k >= 0; --k) {
                    const int idx = (S.head + k) % M;
                    double a = 0;
#pragma unroll
                    for (int i = 0; i < N; ++i) a += S.Sm[idx][i] * q[i];
                    a *= S.rho[idx];
                    S.al[k] = a;
#pragma unroll
                    for (int i = 0; i < N; ++i) q[i] -= a * S.Ym[idx][i];
                }
#pragma unroll
                for (int i = 0; i < N; ++i) q[i] /= S.theta;
                for (int k = 0; k < S.col; ++k) {
                    const int idx = (S.head + k) % M;
                    double b = 0;
#pragma unroll
                    for (int i = 0; i < N; ++i) b += S.Ym[idx][i] * q[i];
                    b *= S.rho[idx];
#pragma unroll
                    for (int i = 0; i < N; ++i) q[i] += (S.al[k] - b) * S.Sm[idx][i];
                }
#pragma unroll
                for (int i = 0; i < N; ++i) d[i] = -q[i];
            }
            // ---- line search (lnsrlb): first trial step
            double dtd = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) dtd += d[i] * d[i];
            const double dnorm = sqrt(dtd);
            S.stp = (S.n_iter == 0) ? fmin(1.0 / dnorm, stpmx) : 1.0;
#pragma unroll
            for (int i = 0; i < N; ++i) { S.t[i] = x[i]; S.r[i] = g[i]; S.d[i] = d[i]; }
            S.fold = f;
            S.ifun = 0;
            S.first = 1;
            in_search = true;
        }
        // ---- one pass of the line-search loop at the current (f, g)
        double gd = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) gd += g[i] * d[i];
        bool ls_fail = false, ls_done = false;
        if (S.ifun == 0) {
            S.gdold = gd;
            if (gd >= 0.0) ls_fail = true;                     // ascent direction: info = -4
        }
        if (!ls_fail) {
            double stp = S.stp;
            Dcsrch ls;
            dcsrch_assign(ls, S.ls);
            const int task = dcsrch(f, gd, stp, 1e-3, 0.9, 0.1, 0.0, stpmx, S.first != 0, ls);
            dcsrch_assign(S.ls, ls);
            S.stp = stp;
            S.first = 0;
            if (task == LS_ERROR) ls_fail = true;
            else if (task == LS_CONV || task == LS_WARN) ls_done = true;
            else {
                ++S.ifun;
                if (S.ifun - 1 >= S.maxls) ls_fail = true;     // checked by mainlb after the return
                else {
#pragma unroll
                    for (int i = 0; i < N; ++i) S.x[i] = stp * d[i] + S.t[i];
                    S.phase = 1;
                    return LB_EVAL;
                }
            }
        }
        in_search = false;
        if (ls_fail) {
#pragma unroll
            for (int i = 0; i < N; ++i) { x[i] = S.t[i]; g[i] = S.r[i]; S.x[i] = x[i]; S.g[i] = g[i]; }
            f = S.fold;
            S.f = f;
            if (S.col == 0) return LB_ABNORMAL;                // ABNORMAL_TERMINATION_IN_LNSRCH
            S.col = 0; S.head = 0; S.theta = 1.0;              // refresh the memory and restart
            continue;
        }
        (void)ls_done;
        // ---- new iterate
        ++S.n_iter;
        double sbgnrm = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) sbgnrm = fmax(sbgnrm, fabs(g[i]));
        if (sbgnrm <= S.pgtol) return LB_CONVERGED_PG;
        const double ddum0 = max3(fabs(S.fold), fabs(f), 1.0);
        if ((S.fold - f) <= LB_EPS * S.factr * ddum0) return LB_CONVERGED_F;
        if (S.n_iter >= S.maxiter) return LB_MAXITER;          // scipy driver: STOP at NEW_X
        // ---- BFGS update
        double rr = 0, dr, ddum, r[N];
#pragma unroll
        for (int i = 0; i < N; ++i) { r[i] = g[i] - S.r[i]; rr += r[i] * r[i]; }
        const double stp = S.stp;
        if (stp == 1.0) {
            dr = gd - S.gdold;
            ddum = -S.gdold;
        } else {
            dr = (gd - S.gdold) * stp;
#pragma unroll
            for (int i = 0; i < N; ++i) d[i] *= stp;
            ddum = -S.gdold * stp;
        }
        if (dr <= LB_EPS * ddum) continue;                     // skip the update
        int slot;
        if (S.col < M) { slot = (S.head + S.col) % M; ++S.col; }
        else { slot = S.head; S.head = (S.head + 1) % M; }
#pragma unroll
        for (int i = 0; i < N; ++i) { S.Sm[slot][i] = d[i]; S.Ym[slot][i] = r[i]; }
        S.rho[slot] = 1.0 / dr;
        S.theta = rr / dr;
    }
}

// Minimise over N variables with a caller-supplied state (wave/block-shared storage on the GPU).
// `eval(x, f, g)` is called by every lane with uniform arguments and must return uniform results;
// every lane advances the state redundantly (same values), so no fence is needed between lanes.
// Returns the stop reason; x, f hold the final iterate.
template <int N, int M, class Eval>
LCFE_FN int lbfgsb_minimize(double x[N], double& f, Eval&& eval, int maxiter, double factr, double pgtol,
                            int maxls, int& n_iter, int& n_eval, LbState<N, M>& S) {
    lb_start(S, x, maxiter, factr, pgtol, maxls);
    int why;
    do {
        double xx[N], ff, gg[N];
        for (int i = 0; i < N; ++i) xx[i] = S.x[i];
        eval(xx, ff, gg);
        S.f = ff;
        for (int i = 0; i < N; ++i) S.g[i] = gg[i];
        why = lb_advance(S);
    } while (why == LB_EVAL);
    for (int i = 0; i < N; ++i) x[i] = S.x[i];
    f = S.f;
    n_iter = S.n_iter;
    n_eval = S.n_eval;
    return why;
}

}  // namespace lcfe
