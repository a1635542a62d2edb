"""csrc/lbfgsb_box.hpp (bounded L-BFGS-B, groundwork for the per-band scikit-learn GP) against
scipy.optimize.minimize(method='L-BFGS-B') itself: the host build of the template is driven with the
SAME Python objective scipy gets, evaluation by evaluation."""
import ctypes

import numpy as np
import pytest
from scipy.optimize import minimize

import hostsim_lib

FG = ctypes.CFUNCTYPE(None, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double))


def run_box(fun, x0, bounds, maxiter=15000, maxfun=15000):
    lib = hostsim_lib.lib()
    lib.hostsim_lbfgsb_box3.restype = ctypes.c_int
    evals = []

    def cb(xp, fp, gp):
        x = np.array([xp[0], xp[1], xp[2]])
        f, g = fun(x)
        evals.append(x)
        fp[0] = f
        for i in range(3):
            gp[i] = g[i]

    x = np.array(x0, float)
    lo = np.array([b[0] for b in bounds], float)
    hi = np.array([b[1] for b in bounds], float)
    f = ctypes.c_double()
    nit, nev = ctypes.c_int(), ctypes.c_int()
    p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    rc = lib.hostsim_lbfgsb_box3(p(x), p(lo), p(hi), FG(cb), maxiter, maxfun, ctypes.c_double(1e7), ctypes.c_double(1e-5),
                                 ctypes.byref(f), ctypes.byref(nit), ctypes.byref(nev), None, 0)
    return x, f.value, nit.value, nev.value, rc, np.array(evals)


def run_scipy(fun, x0, bounds):
    evals = []

    def wrapped(x):
        evals.append(np.array(x))
        return fun(x)

    res = minimize(wrapped, np.array(x0, float), method="L-BFGS-B", jac=True, bounds=bounds)
    return res, np.array(evals)


def quad(x):                       # strictly convex, minimiser outside the box in two coordinates
    A = np.array([[4.0, 1.0, 0.5], [1.0, 3.0, 0.2], [0.5, 0.2, 2.0]])
    c = np.array([3.0, -4.0, 0.5])
    return 0.5 * x @ A @ x - c @ x, A @ x - c


def rosen3(x):
    f = 100 * (x[1] - x[0] ** 2) ** 2 + (1 - x[0]) ** 2 + 100 * (x[2] - x[1] ** 2) ** 2 + (1 - x[1]) ** 2
    g = np.array([-400 * x[0] * (x[1] - x[0] ** 2) - 2 * (1 - x[0]),
                  200 * (x[1] - x[0] ** 2) - 400 * x[1] * (x[2] - x[1] ** 2) - 2 * (1 - x[1]),
                  200 * (x[2] - x[1] ** 2)])
    return f, g


CASES = [
    ("quad_active", quad, [0.5, 0.5, 0.5], [(0.0, 0.4), (-0.5, 1.0), (-1.0, 1.0)]),
    ("quad_interior", quad, [0.0, 0.0, 0.0], [(-10.0, 10.0)] * 3),
    ("rosen_box", rosen3, [-1.2, 1.0, 0.7], [(-2.0, 0.8), (-2.0, 2.0), (-2.0, 2.0)]),
    ("rosen_wide", rosen3, [-1.2, 1.0, 0.7], [(-5.0, 5.0)] * 3),
    ("start_outside", quad, [5.0, -3.0, 9.0], [(0.0, 1.0), (0.0, 1.0), (0.0, 1.0)]),
]


@pytest.mark.parametrize("name,fun,x0,bounds", CASES)
def test_iterates_follow_scipy(name, fun, x0, bounds):
    res, ev_s = run_scipy(fun, x0, bounds)
    x, f, nit, nev, rc, ev_m = run_box(fun, x0, bounds)
    assert nev == res.nfev and nit == res.nit, (name, nev, res.nfev, nit, res.nit)
    # every trial point scipy evaluated is the point this implementation evaluated (the dense form of the
    # limited-memory matrix differs from scipy's compact form by rounding only; on Rosenbrock that drifts
    # to 1e-8 over 60 evaluations and is back to 1e-13 at the minimiser)
    assert np.allclose(ev_m, ev_s, rtol=1e-6, atol=1e-7), name
    assert np.allclose(x, res.x, rtol=1e-9, atol=1e-11) and abs(f - res.fun) <= 1e-11 * max(1.0, abs(res.fun))


def gp_objective(seed, n):
    """-log marginal likelihood of scikit-learn's C * RBF + White kernel in log-parameters, with gradient
    (what GaussianProcessRegressor.fit hands to L-BFGS-B; gaussian_process.py:91-110 of the reference)."""
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, WhiteKernel
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0, 1, n))
    y = np.exp(-0.5 * ((t - 0.4) / 0.15) ** 2) * 2 + rng.normal(0, 0.3, n)
    y = (y - y.mean()) / y.std()
    kernel = ConstantKernel(1.0, (0.01, 100.0)) * RBF(0.2, (0.01, 2.0)) + WhiteKernel(0.1, (1e-5, 10.0))
    gp = GaussianProcessRegressor(kernel=kernel, alpha=np.full(n, 0.05), optimizer=None).fit(t[:, None], y)

    def fun(theta):
        lml, grad = gp.log_marginal_likelihood(theta, eval_gradient=True, clone_kernel=False)
        return -lml, -grad

    return fun, gp.kernel_.theta.copy(), gp.kernel_.bounds.copy()


@pytest.mark.parametrize("seed,n", [(1, 25), (2, 40), (3, 12), (4, 60)])
def test_sklearn_gp_objective_same_optimum_as_scipy(seed, n):
    fun, theta0, bounds = gp_objective(seed, n)
    starts = [theta0] + [np.random.RandomState(42).uniform(bounds[:, 0], bounds[:, 1]) for _ in range(1)]
    for x0 in starts:
        res, ev_s = run_scipy(fun, x0, [tuple(b) for b in bounds])
        x, f, nit, nev, rc, ev_m = run_box(fun, x0, [tuple(b) for b in bounds])
        assert abs(f - res.fun) <= 1e-7 * max(1.0, abs(res.fun)), (seed, f, res.fun)
        assert np.allclose(x, res.x, rtol=1e-4, atol=1e-5), (seed, x, res.x)
        assert abs(nev - res.nfev) <= 2 and nit == res.nit, (seed, nev, res.nfev, nit, res.nit)
