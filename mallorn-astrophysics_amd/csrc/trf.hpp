// trf.hpp -- bounded non-linear least squares: scipy's Trust-Region-Reflective algorithm as the
// reference reaches it through ``curve_fit(..., bounds=...)`` (bazin_fitting.py:128-137,
// train_v55_powerlaw.py:172-184).
//
// Restated from scipy 1.15.3 (line references: scipy/optimize/_lsq/{least_squares,trf,common}.py
// and scipy/optimize/_numdiff.py; digest in SURVEY.md Appendix A):
//   * residual r(x) = w * (model(t; x) - y), w = 1/sigma                (_minpack_py.py:978-996)
//   * x0 -> make_strictly_feasible(rstep=1e-10)                         (least_squares.py:828)
//   * 2-point finite-difference Jacobian, h = sqrt(eps)*sign(x)*max(1,|x|), one-sided bound
//     adjustment, dx recomputed as (x+h)-x                              (_numdiff.py:13-64,146-179,584-596)
//   * trf_bounds main loop, Coleman-Li scaling, select_step (interior / reflected / Cauchy),
//     radius update and termination tests                               (trf.py:128-398, common.py)
//   * exact trust-region sub-problem from an SVD (More' iteration)      (common.py:57-168)
//
// MI355X mapping: one fit per wavefront.  The m rows (data points) are spread over the 64 lanes;
// the n <= 5 parameters and all trust-region scalars are wave-uniform registers.  scipy takes an
// SVD of the (m+n) x n augmented matrix [J*d ; diag(sqrt(diag_h))]; here that matrix lives in LDS
// column-major, is reduced in place to its n x n triangular factor R by Householder reflections
// (the only O(m) linear algebra: batched wave reductions), and R (registers) gets a one-sided
// Jacobi SVD.  Since ||A s|| = ||R s||, every quadratic model evaluation of select_step uses R,
// so the Jacobian is never re-read.  No MFMA: n = 2..5.
#pragma once
#include "wave.hpp"

namespace lcfe {

constexpr double TRF_EPS = 2.220446049250313e-16;
constexpr double TRF_SQRT_EPS = 1.4901161193847656e-08;

// scipy termination codes (>0 success), 0 = max_nfev reached (curve_fit raises -> NaN row),
// negative = the exceptions the reference's try/except turns into NaN.
enum {
    TRF_FAIL_MAXFEV = 0,
    TRF_FAIL_BOUNDS = -1,        // lb >= ub                      least_squares.py:816-818
    TRF_FAIL_X0 = -2,            // x0 outside bounds             least_squares.py:820-821
    TRF_FAIL_NONFINITE = -3,     // non-finite data / f0 / J      _minpack_py.py:930,938; least_squares.py:844-845; svd check_finite
    TRF_FAIL_GEOMETRY = -4,      // intersect_trust_region ValueError (common.py:37-44)
    TRF_FAIL_TOO_FEW = -5,       // not attempted (too few points)
};

LCFE_FN bool finite_d(double x) { return fabs(x) <= 1.79769313486231570815e+308; }   // false for NaN/inf

LCFE_FN double nextafter_toward(double x, double toward) {
    // np.nextafter for finite x != toward
    if (x == toward) return x;
    if (x == 0.0) return (toward > 0) ? 4.9406564584124654e-324 : -4.9406564584124654e-324;
    long long u = __builtin_bit_cast(long long, x);
    const bool up = (toward > x);
    if ((x > 0) == up) u += 1; else u -= 1;
    return __builtin_bit_cast(double, u);
}

template <int N>
struct Vec {
    double v[N];
    LCFE_FN double& operator[](int i) { return v[i]; }
    LCFE_FN const double& operator[](int i) const { return v[i]; }
};

template <int N>
LCFE_FN double vnorm(const Vec<N>& a) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) s += a[i] * a[i];
    return sqrt(s);
}
template <int N>
LCFE_FN double vdot(const Vec<N>& a, const Vec<N>& b) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) s += a[i] * b[i];
    return s;
}

// common.py:401-464 make_strictly_feasible
template <int N>
LCFE_FN void make_strictly_feasible(Vec<N>& x, const Vec<N>& lb, const Vec<N>& ub, double rstep) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double xn = x[i];
        int active = 0;
        if (rstep == 0) {
            if (x[i] <= lb[i]) active = -1;
            if (x[i] >= ub[i]) active = 1;
        } else {
            const double ld = x[i] - lb[i], ud = ub[i] - x[i];
            const double lt = rstep * fmax(1.0, fabs(lb[i])), ut = rstep * fmax(1.0, fabs(ub[i]));
            if (finite_d(lb[i]) && ld <= fmin(ud, lt)) active = -1;
            if (finite_d(ub[i]) && ud <= fmin(ld, ut)) active = 1;
        }
        if (rstep == 0) {
            if (active == -1) xn = nextafter_toward(lb[i], ub[i]);
            if (active == 1) xn = nextafter_toward(ub[i], lb[i]);
        } else {
            if (active == -1) xn = lb[i] + rstep * fmax(1.0, fabs(lb[i]));
            if (active == 1) xn = ub[i] - rstep * fmax(1.0, fabs(ub[i]));
        }
        if (xn < lb[i] || xn > ub[i]) xn = 0.5 * (lb[i] + ub[i]);
        x[i] = xn;
    }
}

// common.py:372-398 step_size_to_bound; hits[i] in {-1,0,1}
template <int N>
LCFE_FN double step_size_to_bound(const Vec<N>& x, const Vec<N>& s, const Vec<N>& lb, const Vec<N>& ub,
                                  int* hits) {
    double steps[N];
    double mn = __builtin_inf();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        steps[i] = __builtin_inf();
        if (s[i] != 0) steps[i] = fmax((lb[i] - x[i]) / s[i], (ub[i] - x[i]) / s[i]);
        // np.min propagates NaN; NaN cannot arise here from finite x, bounds and non-zero s
        mn = (steps[i] < mn) ? steps[i] : mn;
    }
    if (hits) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int sg = (s[i] > 0) - (s[i] < 0);
            hits[i] = (steps[i] == mn) ? sg : 0;
        }
    }
    return mn;
}

// common.py:305-322 minimize_quadratic_1d: min of t*(a*t+b)+c over {lb, ub, interior extremum}
LCFE_FN void minimize_quadratic_1d(double a, double b, double lo, double hi, double c, double& t_out,
                                   double& y_out) {
    double t[3] = {lo, hi, 0};
    int cnt = 2;
    if (a != 0) {
        const double ex = -0.5 * b / a;
        if (lo < ex && ex < hi) { t[2] = ex; cnt = 3; }
    }
    // np.argmin: first minimum, NaN counts as minimum
    int best = 0;
    double yb = t[0] * (a * t[0] + b) + c;
    for (int k = 1; k < cnt; ++k) {
        const double y = t[k] * (a * t[k] + b) + c;
        if (!is_nan(yb) && (is_nan(y) || y < yb)) { yb = y; best = k; }
    }
    t_out = t[best];
    y_out = yb;
}

// Upper-triangular factor of the augmented matrix + everything derived from it.
template <int N>
struct TriFactor {
    double R[N][N];     // R[i][j], j >= i
    LCFE_FN void mul(const Vec<N>& s, Vec<N>& out) const {      // out = R s
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double a = 0;
#pragma unroll
            for (int j = i; j < N; ++j) a += R[i][j] * s[j];
            out[i] = a;
        }
    }
};

// Jacobi rotation (cs, sn) that orthogonalises two columns with squared norms a, b and inner product c (c != 0):
//   zeta = (b - a) / (2 c),  t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)),  cs = 1 / sqrt(1 + t^2),  sn = cs t.
// Three divisions and two square roots -- a third of the instructions of a rotation when they are IEEE sequences
// (11 / 15 instructions each).  On the device they are the hardware reciprocal / reciprocal square root refined by two
// Newton steps (within an ulp or two): a rotation angle that is off in the last bits still orthogonalises the columns to
// working precision, the sweeps converge as before, and the singular values are taken from the column norms afterwards.
#if defined(__HIPCC__)
__device__ __forceinline__ double trf_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    return fma(fma(-d, r, 1.0), r, r);
}
__device__ __forceinline__ double trf_rsqrt(double x) {              // x in [1, 1e300]
    double y = __builtin_amdgcn_rsq(x);
    y = fma(fma(-x * y, 0.5 * y, 0.5), y, y);
    return fma(fma(-x * y, 0.5 * y, 0.5), y, y);
}
__device__ __forceinline__ void jacobi_rotation(double a, double b, double c, double& cs, double& sn) {
    const double zeta = (b - a) * trf_rcp(2.0 * c);
    const double az = fabs(zeta), w = fma(zeta, zeta, 1.0);
    const double h = (w < 1e300) ? w * trf_rsqrt(w) : az;           // sqrt(1 + zeta^2); |zeta| beyond 1e150
    const double den = az + h;
    const double t = (den < 1e300) ? trf_rcp(den) : 0.0;
    const double tt = (zeta >= 0) ? t : -t;
    cs = trf_rsqrt(fma(tt, tt, 1.0));
    sn = cs * tt;
}
#else
inline void jacobi_rotation(double a, double b, double c, double& cs, double& sn) {
    const double zeta = (b - a) / (2.0 * c);
    const double tt = ((zeta >= 0) ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    cs = 1.0 / sqrt(1.0 + tt * tt);
    sn = cs * tt;
}
#endif

// One-sided Jacobi SVD of an N x N matrix given by columns W (W = R initially): on exit
// sv[j] = singular values (unordered), V = right vectors (columns), and ut[j] = u_j . q for the
// vector q (the Q^T f part), which is all the trust-region solver needs of U.
template <int N>
LCFE_FN void jacobi_svd(const TriFactor<N>& T, const Vec<N>& q, double sv[N], double V[N][N], double ut[N]) {
    double W[N][N];     // W[i][j] row i, column j
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            W[i][j] = (j >= i) ? T.R[i][j] : 0.0;
            V[i][j] = (i == j) ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 40; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < N - 1; ++p) {
#pragma unroll
            for (int r = p + 1; r < N; ++r) {
                double a = 0, b = 0, c = 0;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    a += W[i][p] * W[i][p];
                    b += W[i][r] * W[i][r];
                    c += W[i][p] * W[i][r];
                }
                if (c != 0.0 && c * c > 1e-30 * (a * b)) {       // |c| > 1e-15 sqrt(a b), without the square root
                    rotated = true;
                    double cs, sn;
                    jacobi_rotation(a, b, c, cs, sn);
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        const double wp = W[i][p], wr = W[i][r];
                        W[i][p] = cs * wp - sn * wr;
                        W[i][r] = sn * wp + cs * wr;
                        const double vp = V[i][p], vr = V[i][r];
                        V[i][p] = cs * vp - sn * vr;
                        V[i][r] = sn * vp + cs * vr;
                    }
                }
            }
        }
        if (!rotated) break;
#ifdef LCFE_TRF_TRACE
        if (sweep >= 8) printf("   jacobi sweep %d\n", sweep);
#endif
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double s2 = 0, uq = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) { s2 += W[i][j] * W[i][j]; uq += W[i][j] * q[i]; }
        const double s = sqrt(s2);
        sv[j] = s;
        ut[j] = (s > 0) ? uq / s : 0.0;
    }
}

// common.py:57-168 solve_lsq_trust_region with the SVD pieces above (sv unordered).
template <int N>
LCFE_FN void solve_lsq_trust_region(int m, const double uf[N], const double sv[N], const double V[N][N],
                                    double Delta, double& alpha, Vec<N>& p) {
    double suf[N];
    double smax = 0, smin = __builtin_inf();
#pragma unroll
    for (int j = 0; j < N; ++j) {
        suf[j] = sv[j] * uf[j];
        smax = fmax(smax, sv[j]);
        smin = fmin(smin, sv[j]);
    }
    const bool full_rank = (m >= N) && (smin > TRF_EPS * m * smax);
    if (full_rank) {
        double nn = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double a = 0;
#pragma unroll
            for (int j = 0; j < N; ++j) a += V[i][j] * (uf[j] / sv[j]);
            p[i] = -a;
            nn += a * a;
        }
        if (sqrt(nn) <= Delta) { alpha = 0.0; return; }
    }
    double nsuf = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) nsuf += suf[j] * suf[j];
    double alpha_upper = sqrt(nsuf) / Delta;
    auto phi_and_derivative = [&](double al, double& phi, double& phi_prime) {
        double pn2 = 0, sp = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const double den = sv[j] * sv[j] + al;
            const double q = suf[j] / den;
            pn2 += q * q;
            sp += suf[j] * suf[j] / (den * den * den);
        }
        const double pn = sqrt(pn2);
        phi = pn - Delta;
        phi_prime = -sp / pn;
    };
    double alpha_lower = 0.0;
    if (full_rank) {
        double phi, phip;
        phi_and_derivative(0.0, phi, phip);
        alpha_lower = -phi / phip;
    }
    if (!full_rank && alpha == 0.0) alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
    for (int it = 0; it < 10; ++it) {
        if (alpha < alpha_lower || alpha > alpha_upper)
            alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
        double phi, phip;
        phi_and_derivative(alpha, phi, phip);
        if (phi < 0) alpha_upper = alpha;
        const double ratio = phi / phip;
        alpha_lower = fmax(alpha_lower, alpha - ratio);
        alpha -= (phi + Delta) * ratio / Delta;
        if (fabs(phi) < 0.01 * Delta) break;
    }
    double nn = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double a = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) a += V[i][j] * (suf[j] / (sv[j] * sv[j] + alpha));
        p[i] = -a;
        nn += a * a;
    }
    const double sc = Delta / sqrt(nn);
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] *= sc;
}

// quadratic model pieces through R:  q(s) = 0.5*||R s||^2 + g_h . s
template <int N>
LCFE_FN double evaluate_quadratic(const TriFactor<N>& T, const Vec<N>& g_h, const Vec<N>& s) {
    Vec<N> rs;
    T.mul(s, rs);
    return 0.5 * vdot(rs, rs) + vdot(s, g_h);
}

// trf.py:128-202 select_step.  Returns false on the ValueError cases of intersect_trust_region.
template <int N>
LCFE_FN bool select_step(const Vec<N>& x, const TriFactor<N>& T, const Vec<N>& g_h, Vec<N> p, Vec<N> p_h,
                         const Vec<N>& d, double Delta, const Vec<N>& lb, const Vec<N>& ub, double theta,
                         Vec<N>& step, Vec<N>& step_h, double& predicted) {
    bool inb = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double xp = x[i] + p[i];
        inb = inb && (xp >= lb[i]) && (xp <= ub[i]);
    }
    if (inb) {
        step = p;
        step_h = p_h;
        predicted = -evaluate_quadratic(T, g_h, p_h);
        return true;
    }
    int hits[N];
    const double p_stride = step_size_to_bound(x, p, lb, ub, hits);
    Vec<N> r_h, r, x_on_bound;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        r_h[i] = hits[i] ? -p_h[i] : p_h[i];
        r[i] = d[i] * r_h[i];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        p[i] *= p_stride;
        p_h[i] *= p_stride;
        x_on_bound[i] = x[i] + p[i];
    }
    // intersect_trust_region(p_h, r_h, Delta) -> positive root   (common.py:18-54)
    double to_tr;
    {
        const double a = vdot(r_h, r_h);
        if (a == 0) return false;
        const double b = vdot(p_h, r_h);
        const double c = vdot(p_h, p_h) - Delta * Delta;
        if (c > 0) return false;
        const double dd = sqrt(b * b - a * c);
        const double q = -(b + copysign(dd, b));
        const double t1 = q / a, t2 = c / q;
        to_tr = (t1 < t2) ? t2 : t1;
    }
    const double to_bound = step_size_to_bound(x_on_bound, r, lb, ub, (int*)nullptr);
    double r_stride = fmin(to_bound, to_tr);
    double r_stride_l, r_stride_u;
    if (r_stride > 0) {
        r_stride_l = (1 - theta) * p_stride / r_stride;
        r_stride_u = (r_stride == to_bound) ? theta * to_bound : to_tr;
    } else {
        r_stride_l = 0;
        r_stride_u = -1;
    }
    double r_value;
    if (r_stride_l <= r_stride_u) {
        // build_quadratic_1d(J_h, g_h, r_h, s0=p_h, diag=diag_h) through R
        Vec<N> v, u;
        T.mul(r_h, v);
        T.mul(p_h, u);
        const double a = 0.5 * vdot(v, v);
        const double b = vdot(g_h, r_h) + vdot(u, v);
        const double c = 0.5 * vdot(u, u) + vdot(g_h, p_h);
        minimize_quadratic_1d(a, b, r_stride_l, r_stride_u, c, r_stride, r_value);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            r_h[i] = r_h[i] * r_stride + p_h[i];
            r[i] = r_h[i] * d[i];
        }
    } else {
        r_value = __builtin_inf();
    }
#pragma unroll
    for (int i = 0; i < N; ++i) { p[i] *= theta; p_h[i] *= theta; }
    const double p_value = evaluate_quadratic(T, g_h, p_h);
    Vec<N> ag_h, ag;
#pragma unroll
    for (int i = 0; i < N; ++i) { ag_h[i] = -g_h[i]; ag[i] = d[i] * ag_h[i]; }
    const double to_tr2 = Delta / vnorm(ag_h);
    const double to_bound2 = step_size_to_bound(x, ag, lb, ub, (int*)nullptr);
    double ag_stride = (to_bound2 < to_tr2) ? theta * to_bound2 : to_tr2;
    double ag_value;
    {
        Vec<N> v;
        T.mul(ag_h, v);
        const double a = 0.5 * vdot(v, v);
        const double b = vdot(g_h, ag_h);
        minimize_quadratic_1d(a, b, 0.0, ag_stride, 0.0, ag_stride, ag_value);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) { ag_h[i] *= ag_stride; ag[i] *= ag_stride; }
    if (p_value < r_value && p_value < ag_value) {
        step = p; step_h = p_h; predicted = -p_value;
    } else if (r_value < p_value && r_value < ag_value) {
        step = r; step_h = r_h; predicted = -r_value;
    } else {
        step = ag; step_h = ag_h; predicted = -ag_value;
    }
    return true;
}

// LDS working set of one fit of up to CAP rows and N parameters.
template <int N, int CAP>
struct TrfLds {
    double A[N + 1][CAP + N];   // columns of [J*d ; diag] plus the residual column (column-major)
    double r[CAP];              // residual at x
    double rn[CAP];             // residual at the trial point
    double w[CAP];              // 1/sigma (1 for unweighted fits)
};

struct TrfResult {
    int status;
    int nfev;
};

#if defined(LCFE_TRF_PROF) && defined(__HIPCC__)
__device__ unsigned long long g_trf_prof[8];
#define TRF_T0() unsigned long long tt0__ = __builtin_readcyclecounter()
#define TRF_T(slot_) do { unsigned long long tt1__ = __builtin_readcyclecounter(); if (W::lane() == 0) atomicAdd(&g_trf_prof[slot_], tt1__ - tt0__); tt0__ = tt1__; } while (0)
#else
#define TRF_T0() do {} while (0)
#define TRF_T(slot_) do {} while (0)
#endif

// Model concept:  static constexpr int NP;  LCFE_FN double operator()(double t, const Vec<NP>& p) const
//
// Fits  r_i(x) = w_i*(model(t_i;x) - y_i), i < m.  t, y are wave-shared arrays (LDS); S.w must hold
// the weights.  x holds p0 on entry and the solution on exit (wave-uniform).
//
// Store: anything with S.A[k][i] (k <= NP columns of m + NP rows), S.r[i], S.rn[i], S.w[i] -- a
// TrfLds block, or a TrfView of pointers into a pool shared by the bands of one light curve.
template <int N>
struct TrfView {
    double* A[N + 1];
    double* r;
    double* rn;
    double* w;
};

// ---- the solver as a resumable machine.  One fit is a TrfState plus its Store; trf_begin runs the prologue, then
// trf_outer (scaling, termination tests, QR + SVD of the augmented Jacobian) and trf_inner (ONE trial step; on the
// end of scipy's inner loop also the acceptance and the new Jacobian) alternate until phase == TRF_PH_DONE.  The
// arithmetic and its order are those of the monolithic loop this replaces; what changes is that a caller can
// interleave the phases of several fits in one flat loop (lcfe.hip: the fit kernels keep every lane group busy by
// handing a finished group its next fit while its neighbours iterate on).
enum { TRF_PH_OUTER = 0, TRF_PH_INNER = 1, TRF_PH_DONE = 2 };

template <int N>
struct TrfState {
    Vec<N> x, lb, ub, g, v, dv, d, diag_h, g_h, x_new;
    TriFactor<N> T;
    double sv[N], Vm[N][N], uf[N];
    double cost, cost_new, Delta, alpha, theta, actual;
    int status;                // -99 while running (scipy's None)
    int phase;
    int max_nfev;
    TrfResult res;
};

// how many terms of its value a model keeps from the residual evaluation at x for the Jacobian at x (0: none)
template <class Model, class = void>
struct trf_shared_terms { static constexpr int value = 0; };
template <class Model>
struct trf_shared_terms<Model, decltype((void)Model::kSharedTerms)> { static constexpr int value = Model::kSharedTerms; };

template <class W, class Model, class Store>
LCFE_FN void trf_residual(const Model& model, const double* t, const double* y, int m, Store& S, const Vec<Model::NP>& xx,
                          double* out, double& cost2, bool& finite) {
    double c = 0;
    bool ok = true;
    for (int i = W::lane(); i < m; i += W::LANES) {
        double mv;
        if constexpr (trf_shared_terms<Model>::value > 0) {
            // the terms of the model at xx stay in the first matrix columns (free between the QR and the next Jacobian)
            // for the Jacobian that follows an accepted point; the value is operator()'s, term by term
            double q[trf_shared_terms<Model>::value];
            model.terms(t[i], xx, q);
#pragma unroll
            for (int k = 0; k < trf_shared_terms<Model>::value; ++k) S.A[k][i] = q[k];
            mv = model.value(xx, q);
        } else {
            mv = model(t[i], xx);
        }
        const double v = S.w[i] * (mv - y[i]);
        out[i] = v;
        ok = ok && finite_d(v);
        c += v * v;
    }
    cost2 = W::sum(c);
    finite = W::all(ok);
}

// _numdiff.py:146-179 absolute step of component k; :13-64 one-sided bound adjustment
template <int N>
LCFE_FN double trf_fd_step(const Vec<N>& xx, const Vec<N>& lb, const Vec<N>& ub, int k) {
    double h = TRF_SQRT_EPS * ((xx[k] >= 0) ? 1.0 : -1.0) * fmax(1.0, fabs(xx[k]));
    const double lower = xx[k] - lb[k], upper = ub[k] - xx[k];
    const double xt = xx[k] + h;
    const bool violated = (xt < lb[k]) || (xt > ub[k]);
    const bool fitting = fabs(h) <= fmax(lower, upper);
    if (violated && fitting) h = -h;
    if (!fitting) h = (upper >= lower) ? upper : -lower;
    return h;
}

// one row of the FD Jacobian, components K .. N-1
template <class Model, class Store, int N, int K, int NT>
LCFE_FN void trf_jacobian_row(const Model& model, double ti, double yi, double wi, double ri, int i, const Vec<N> (&x1)[N],
                              const double (&dx)[N], const double (&at_x)[NT], Store& S, double (&gacc)[N], bool& ok) {
    if constexpr (K < N) {
        const double f1 = wi * (model.template stepped<K>(ti, x1[K], at_x) - yi);
        const double jv = (f1 - ri) / dx[K];
        S.A[K][i] = jv;
        ok = ok && finite_d(jv);
        gacc[K] += jv * ri;
        trf_jacobian_row<Model, Store, N, K + 1>(model, ti, yi, wi, ri, i, x1, dx, at_x, S, gacc, ok);
    }
}

// FD Jacobian into S.A columns (unscaled), returns g = J^T f and a finiteness flag
template <class W, class Model, class Store>
LCFE_FN void trf_jacobian(const Model& model, const double* t, const double* y, int m, Store& S, const Vec<Model::NP>& xx,
                          const Vec<Model::NP>& lb, const Vec<Model::NP>& ub, Vec<Model::NP>& g, bool& finite) {
    constexpr int N = Model::NP;
    const int lane = W::lane();
    bool ok = true;
    double gacc[N];
    if constexpr (trf_shared_terms<Model>::value > 0) {
        // rows outside, components inside: the terms of the model that a step in component k does not touch are the
        // ones the residual evaluation at x left in the first matrix columns (same values, same arithmetic, same
        // order of the sums over the rows).  Every Jacobian follows a residual evaluation at the same x: the start
        // (trf_begin) and an accepted trial point (trf_inner).
        Vec<N> x1[N];
        double dx[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const double h = trf_fd_step<N>(xx, lb, ub, k);
            x1[k] = xx;
            x1[k][k] = xx[k] + h;
            dx[k] = x1[k][k] - xx[k];
            gacc[k] = 0;
        }
        for (int i = lane; i < m; i += W::LANES) {
            const double ti = t[i], yi = y[i], wi = S.w[i], ri = S.r[i];
            double at_x[trf_shared_terms<Model>::value];
#pragma unroll
            for (int k = 0; k < trf_shared_terms<Model>::value; ++k) at_x[k] = S.A[k][i];
            trf_jacobian_row<Model, Store, N, 0>(model, ti, yi, wi, ri, i, x1, dx, at_x, S, gacc, ok);
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const double h = trf_fd_step<N>(xx, lb, ub, k);
            Vec<N> x1 = xx;
            x1[k] = xx[k] + h;
            const double dx = x1[k] - xx[k];
            double ga = 0;
            for (int i = lane; i < m; i += W::LANES) {
                const double f1 = S.w[i] * (model(t[i], x1) - y[i]);
                const double jv = (f1 - S.r[i]) / dx;
                S.A[k][i] = jv;
                ok = ok && finite_d(jv);
                ga += jv * S.r[i];
            }
            gacc[k] = ga;
        }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) g[k] = W::sum(gacc[k]);
    finite = W::all(ok);
}

// CL scaling  (common.py:467-508); all bounds here are finite
template <int N>
LCFE_FN void trf_cl_scaling(TrfState<N>& Z) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        Z.v[i] = 1.0; Z.dv[i] = 0.0;
        if (Z.g[i] < 0 && finite_d(Z.ub[i])) { Z.v[i] = Z.ub[i] - Z.x[i]; Z.dv[i] = -1.0; }
        if (Z.g[i] > 0 && finite_d(Z.lb[i])) { Z.v[i] = Z.x[i] - Z.lb[i]; Z.dv[i] = 1.0; }
    }
}

// prologue of curve_fit / least_squares + first residual and Jacobian.  Z.x, Z.lb, Z.ub, Z.max_nfev set by the caller.
template <class W, class Model, class Store>
LCFE_FN void trf_begin(const Model& model, const double* t, const double* y, int m, TrfState<Model::NP>& Z, Store& S) {
    constexpr int N = Model::NP;
    const int lane = W::lane();
    Z.res = TrfResult{TRF_FAIL_MAXFEV, 0};
    Z.phase = TRF_PH_DONE;
    bool data_ok = true;
    for (int i = lane; i < m; i += W::LANES) data_ok = data_ok && finite_d(t[i]) && finite_d(y[i]);
    if (!W::all(data_ok)) { Z.res.status = TRF_FAIL_NONFINITE; return; }
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (!(Z.lb[i] < Z.ub[i])) { Z.res.status = TRF_FAIL_BOUNDS; return; }     // also catches NaN bounds
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (!(Z.x[i] >= Z.lb[i] && Z.x[i] <= Z.ub[i])) { Z.res.status = TRF_FAIL_X0; return; }
    make_strictly_feasible(Z.x, Z.lb, Z.ub, 1e-10);
    double cost2;
    bool fin;
    trf_residual<W, Model, Store>(model, t, y, m, S, Z.x, S.r, cost2, fin);
    Z.res.nfev = 1;
    if (!fin) { Z.res.status = TRF_FAIL_NONFINITE; return; }
    Z.cost = 0.5 * cost2;
    W::sync();
    trf_jacobian<W, Model, Store>(model, t, y, m, S, Z.x, Z.lb, Z.ub, Z.g, fin);
    if (!fin) { Z.res.status = TRF_FAIL_NONFINITE; return; }
    trf_cl_scaling<N>(Z);
    {
        double s = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) { const double q = Z.x[i] / sqrt(Z.v[i]); s += q * q; }
        Z.Delta = sqrt(s);                                 // trf.py:232-236 (x0 * scale_inv / v**0.5)
        if (Z.Delta == 0) Z.Delta = 1.0;
    }
    Z.alpha = 0.0;
    Z.status = -99;
    Z.phase = TRF_PH_OUTER;
}

// top of scipy's outer loop up to the SVD; ends the fit when a termination test fires
template <class W, class Model, class Store>
LCFE_FN void trf_outer(int m, TrfState<Model::NP>& Z, Store& S) {
    constexpr int N = Model::NP;
    const int lane = W::lane();
    const int M = m + N;
    const double gtol = 1e-8;
    TRF_T0();
    trf_cl_scaling<N>(Z);
    double g_norm = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) g_norm = fmax(g_norm, fabs(Z.g[i] * Z.v[i]));
    if (g_norm < gtol) Z.status = 1;
#ifdef LCFE_TRF_TRACE
    printf("it nfev=%d cost=%.17g Delta=%.17g g_norm=%.6e x=", Z.res.nfev, Z.cost, Z.Delta, g_norm);
    for (int i = 0; i < N; ++i) printf("%.17g ", Z.x[i]);
    printf("\n");
#endif
    if (Z.status != -99 || Z.res.nfev == Z.max_nfev) {
        Z.res.status = (Z.status == -99) ? TRF_FAIL_MAXFEV : Z.status;
        Z.phase = TRF_PH_DONE;
        return;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        Z.d[i] = sqrt(Z.v[i]);
        Z.diag_h[i] = Z.g[i] * Z.dv[i];
        Z.g_h[i] = Z.d[i] * Z.g[i];
    }
    // ---- augmented matrix in LDS: scale J columns by d, append diag rows and residual column
    W::sync();
    for (int i = lane; i < M; i += W::LANES) {
        if (i < m) {
#pragma unroll
            for (int k = 0; k < N; ++k) S.A[k][i] *= Z.d[k];
            S.A[N][i] = S.r[i];
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) S.A[k][i] = (i - m == k) ? sqrt(Z.diag_h[k]) : 0.0;
            S.A[N][i] = 0.0;
        }
    }
    W::sync();
    TRF_T(0);
    // ---- Householder QR (in place); R and Q^T f end in rows 0..N-1
    Vec<N> qtf;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        // norm of column k below (and including) row k, and its pivot element
        double nn = 0;
        for (int i = lane; i < M; i += W::LANES)
            if (i >= k) { const double a = S.A[k][i]; nn += a * a; }
        nn = W::sum(nn);
        const double akk = S.A[k][k];
        const double nrm = sqrt(nn);
        const double alpha_h = (akk > 0) ? -nrm : nrm;
        const double vk = akk - alpha_h;                 // v = x - alpha e_k
        const double vtv = nn - 2.0 * alpha_h * akk + alpha_h * alpha_h;
        // dot products of v with the remaining columns (and the residual column)
        double dots[N + 1];
#pragma unroll
        for (int j = 0; j <= N; ++j) dots[j] = 0;
        if (vtv > 0) {
            for (int i = lane; i < M; i += W::LANES) {
                if (i < k) continue;
                const double vi = (i == k) ? vk : S.A[k][i];
#pragma unroll
                for (int j = 0; j <= N; ++j)
                    if (j > k) dots[j] += vi * S.A[j][i];
            }
#pragma unroll
            for (int j = 0; j <= N; ++j)
                if (j > k) dots[j] = W::sum(dots[j]);
            W::sync();
            for (int i = lane; i < M; i += W::LANES) {
                if (i < k) continue;
                const double vi = (i == k) ? vk : S.A[k][i];
#pragma unroll
                for (int j = 0; j <= N; ++j)
                    if (j > k) S.A[j][i] -= (2.0 * dots[j] / vtv) * vi;
            }
        }
        W::sync();
        Z.T.R[k][k] = (vtv > 0) ? alpha_h : akk;
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (j > k) Z.T.R[k][j] = S.A[j][k];
        qtf[k] = S.A[N][k];
        W::sync();
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j)
            if (j < i) Z.T.R[i][j] = 0.0;
    TRF_T(1);
    jacobi_svd<N>(Z.T, qtf, Z.sv, Z.Vm, Z.uf);
    TRF_T(2);
#ifdef LCFE_TRF_TRACE
    printf("   sv="); for (int i = 0; i < N; ++i) printf("%.17g ", Z.sv[i]);
    printf(" uf="); for (int i = 0; i < N; ++i) printf("%.17g ", Z.uf[i]);
    printf(" qtf="); for (int i = 0; i < N; ++i) printf("%.17g ", qtf[i]);
    printf("\n   R="); for (int i = 0; i < N; ++i) for (int j = i; j < N; ++j) printf("%.17g ", Z.T.R[i][j]);
    printf("\n");
#endif
    Z.theta = fmax(0.995, 1 - g_norm);
    Z.actual = -1.0;
    Z.cost_new = Z.cost;
    Z.phase = TRF_PH_INNER;
}

// one pass of scipy's inner loop (one trial point); when that loop ends, the acceptance step and the new Jacobian
template <class W, class Model, class Store>
LCFE_FN void trf_inner(const Model& model, const double* t, const double* y, int m, TrfState<Model::NP>& Z, Store& S) {
    constexpr int N = Model::NP;
    const int lane = W::lane();
    const double ftol = 1e-8, xtol = 1e-8;
    TRF_T0();
    bool loop_ends = false;
    {
        Vec<N> p_h, p, step, step_h;
        solve_lsq_trust_region<N>(m, Z.uf, Z.sv, Z.Vm, Z.Delta, Z.alpha, p_h);
#pragma unroll
        for (int i = 0; i < N; ++i) p[i] = Z.d[i] * p_h[i];
        double predicted;
        TRF_T(3);
        if (!select_step<N>(Z.x, Z.T, Z.g_h, p, p_h, Z.d, Z.Delta, Z.lb, Z.ub, Z.theta, step, step_h, predicted)) {
            Z.res.status = TRF_FAIL_GEOMETRY;
            Z.phase = TRF_PH_DONE;
            return;
        }
#pragma unroll
        for (int i = 0; i < N; ++i) Z.x_new[i] = Z.x[i] + step[i];
        make_strictly_feasible(Z.x_new, Z.lb, Z.ub, 0.0);
        TRF_T(4);
        double c2;
        bool f_ok;
        trf_residual<W, Model, Store>(model, t, y, m, S, Z.x_new, S.rn, c2, f_ok);
        Z.res.nfev += 1;
        TRF_T(5);
        const double step_h_norm = vnorm(step_h);
        if (!f_ok) Z.Delta = 0.25 * step_h_norm;
        else {
            Z.cost_new = 0.5 * c2;
            Z.actual = Z.cost - Z.cost_new;
            // update_tr_radius (common.py:222-248)
            double ratio;
            if (predicted > 0) ratio = Z.actual / predicted;
            else if (predicted == 0 && Z.actual == 0) ratio = 1;
            else ratio = 0;
            double Delta_new = Z.Delta;
            if (ratio < 0.25) Delta_new = 0.25 * step_h_norm;
            else if (ratio > 0.75 && step_h_norm > 0.95 * Z.Delta) Delta_new = Z.Delta * 2.0;
            // check_termination (common.py:705-717)
            const double step_norm = vnorm(step), x_norm = vnorm(Z.x);
            const bool ftol_ok = (Z.actual < ftol * Z.cost) && (ratio > 0.25);
            const bool xtol_ok = step_norm < xtol * (xtol + x_norm);
            if (ftol_ok && xtol_ok) Z.status = 4;
            else if (ftol_ok) Z.status = 2;
            else if (xtol_ok) Z.status = 3;
            if (Z.status != -99) loop_ends = true;
            else {
                Z.alpha *= Z.Delta / Delta_new;
                Z.Delta = Delta_new;
            }
        }
    }
    // scipy: while actual_reduction <= 0 and nfev < max_nfev
    if (!loop_ends && Z.actual <= 0 && Z.res.nfev < Z.max_nfev) return;       // another trial (phase stays INNER)
    if (Z.actual > 0) {
        Z.x = Z.x_new;
        W::sync();
        for (int i = lane; i < m; i += W::LANES) S.r[i] = S.rn[i];
        Z.cost = Z.cost_new;
        W::sync();
        // scipy recomputes J here even when terminating; curve_fit then takes an SVD of it
        // (check_finite) -> a non-finite final Jacobian is a failure too.
        TRF_T(6);
        bool fin;
        trf_jacobian<W, Model, Store>(model, t, y, m, S, Z.x, Z.lb, Z.ub, Z.g, fin);
        TRF_T(7);
        if (!fin) { Z.res.status = TRF_FAIL_NONFINITE; Z.phase = TRF_PH_DONE; return; }
    }
    Z.phase = TRF_PH_OUTER;
}

// The fit in one call: r_i(x) = w_i*(model(t_i;x) - y_i), i < m.  t, y are wave-shared arrays (LDS); S.w must hold
// the weights.  x holds p0 on entry and the solution on exit (wave-uniform).
template <class W, class Model, class Store>
LCFE_FN TrfResult trf_fit(const Model& model, const double* t, const double* y, int m,
                          Vec<Model::NP>& x, const Vec<Model::NP>& lb, const Vec<Model::NP>& ub, int max_nfev,
                          Store& S) {
    TrfState<Model::NP> Z;
    Z.x = x; Z.lb = lb; Z.ub = ub; Z.max_nfev = max_nfev;
    trf_begin<W, Model, Store>(model, t, y, m, Z, S);
    while (Z.phase != TRF_PH_DONE) {
        if (Z.phase == TRF_PH_OUTER) trf_outer<W, Model, Store>(m, Z, S);
        else trf_inner<W, Model, Store>(model, t, y, m, Z, S);
    }
    x = Z.x;
    return Z.res;
}

}  // namespace lcfe
