"""Oracle for src/features/multiband_gp.py -> 27 columns.   *** PARITY UNPINNED ***

The reference fits the GP with ``george`` (``multiband_gp.py:19-20``; ``requirements.txt:16``
``george>=0.4.0``, unpinned), which is neither installed nor vendored, and no file of the
reference holds outputs of this path.  This module therefore restates george's published
algorithm (george 0.4.x ``GP``/``BasicSolver``/``kernels``) around the reference's own call
sites, and drives the real ``scipy.optimize.minimize(L-BFGS-B)`` exactly as
``multiband_gp.py:158-164`` does.  Every george semantic that cannot be verified offline is a
named switch below:

* ``CONST_DIV_NDIM``: ``float * kernel`` builds ``ConstantKernel(log(float / ndim))``.
* ``PARAM_ORDER``: the unfrozen parameter vector is
  ``[mean:value, kernel:k1:log_constant, kernel:k2:metric:log_M_0_0, ...log_M_1_1]``; the
  reference reads ``params[0..2]`` as (log_amp, log_metric_time, log_metric_wave)
  (``multiband_gp.py:171-180``), i.e. with george's real order it labels
  ``exp(mean)`` / ``sqrt(constant)`` / time-metric -- reproduced faithfully.
"""
import numpy as np
from scipy.linalg import cho_factor, cho_solve
from scipy.optimize import minimize

NCOL = 27
WAVE = np.array([3670.0, 4825.0, 6222.0, 7545.0, 8691.0, 9710.0])     # multiband_gp.py:26-29
TINY = 1.25e-12                                                         # george.gp.TINY
CONST_DIV_NDIM = True
PARAM_ORDER = "mean_first"      # or "kernel_first"
EPOCHS = (0, 20, 50, 100)


def prepare(o):
    """multiband_gp.py:34-87 -> (X, y, yerr, scale) or None."""
    m = (o.b < 6) & ~np.isnan(o.f) & ~np.isnan(o.e) & (o.e > 0)          # :51-59
    if np.sum(m) < 10:                                                   # :66
        return None
    t, f, e = o.t[m], o.f[m], o.e[m]
    lam = WAVE[o.b[m]]
    t = t - t.min()                                                      # :75
    nz = f[f != 0]
    scale = np.median(np.abs(nz)) if nz.size else np.nan                 # :78-80
    if scale == 0:
        scale = 1.0
    return np.column_stack([t, lam]), f / scale, e / scale, scale


class _GP:
    """george.GP(amp * Matern32Kernel(metric, ndim=2), mean=c, fit_mean=True) restated."""

    def __init__(self, X, y, yerr):
        self.X, self.y, self.yerr2 = X, y, yerr ** 2
        self.dt2 = (X[:, None, 0] - X[None, :, 0]) ** 2
        self.dl2 = (X[:, None, 1] - X[None, :, 1]) ** 2

    def kernel(self, p, dt2, dl2):
        c, m0, m1 = np.exp(p[1]), np.exp(p[2]), np.exp(p[3])
        u = np.sqrt(3.0 * (dt2 / m0 + dl2 / m1))
        return c * (1.0 + u) * np.exp(-u), u, c, m0, m1

    def factor(self, p):
        K, u, c, m0, m1 = self.kernel(p, self.dt2, self.dl2)
        Kn = K + np.diag(self.yerr2 + TINY)
        L = cho_factor(Kn, lower=True, overwrite_a=False)
        r = self.y - p[0]
        alpha = cho_solve(L, r)
        return K, u, c, m0, m1, L, r, alpha

    def nll(self, p):
        # multiband_gp.py:141-147 (george GP.log_likelihood, quiet=True)
        try:
            K, u, c, m0, m1, L, r, alpha = self.factor(p)
            ll = -0.5 * (r @ alpha + 2.0 * np.sum(np.log(np.diag(L[0]))) + len(r) * np.log(2 * np.pi))
            return -ll if np.isfinite(ll) else 1e25
        except Exception:
            return 1e25

    def grad_nll(self, p):
        # multiband_gp.py:149-154 (george GP.grad_log_likelihood)
        try:
            K, u, c, m0, m1, L, r, alpha = self.factor(p)
            Kinv = cho_solve(L, np.eye(len(r)))
            A = np.outer(alpha, alpha) - Kinv
            e = 1.5 * c * np.exp(-u)                  # dK/d(log M_k) = e * Delta_k^2 / M_k
            g = np.array([np.sum(alpha),
                          0.5 * np.sum(A * K),
                          0.5 * np.sum(A * e * self.dt2 / m0),
                          0.5 * np.sum(A * e * self.dl2 / m1)])
            return -g
        except Exception:
            return np.zeros_like(p)

    def predict(self, p, xs):
        K, u, c, m0, m1, L, r, alpha = self.factor(p)
        ks, *_ = self.kernel(p, (xs[:, None, 0] - self.X[None, :, 0]) ** 2,
                             (xs[:, None, 1] - self.X[None, :, 1]) ** 2)
        return p[0] + ks @ alpha


def fit(X, y, yerr, info=None):
    """multiband_gp.py:90-193 -> (gp, p_opt, 5 features) ; gp None on failure."""
    feats = np.full(5, np.nan)
    try:
        amp0 = np.var(y)                                                 # :125
        gp = _GP(X, y, yerr)
        c0 = amp0 / 2.0 if CONST_DIV_NDIM else amp0
        p0 = np.array([np.mean(y), np.log(c0), np.log(100.0 ** 2), np.log(6000.0 ** 2)])   # :129-135
        res = minimize(gp.nll, p0, jac=gp.grad_nll, method="L-BFGS-B", options={"maxiter": 100})
        p = res.x
        if info is not None:
            info.update(nit=res.nit, nfev=res.nfev, x=res.x.copy(), fun=res.fun)
        q = p if PARAM_ORDER == "mean_first" else p[[1, 2, 3, 0]]        # :171-176
        amplitude = np.exp(q[0])
        ts = np.sqrt(np.exp(q[1]))
        ws = np.sqrt(np.exp(q[2]))
        feats = np.array([amplitude, ts, ws, -res.fun, ts / (ws / 1000)])   # :182-188
        return gp, p, feats
    except Exception:
        return None, None, feats


def extract_one(o):
    out = np.full(NCOL, np.nan)
    prep = prepare(o)
    if prep is None:                                                     # :306-322
        return out
    X, y, yerr, scale = prep
    with np.errstate(all="ignore"):
        gp, p, feats = fit(X, y, yerr)
        out[:5] = feats
        if gp is None:
            return out
        rmask = o.b == 2                                                 # :331-338
        if rmask.sum() > 0:
            rf, rt = o.f[rmask], o.t[rmask]
            peak_time = rt[np.nanargmax(rf)] - np.min(o.t)
        else:
            peak_time = o.t[np.nanargmax(o.f)] - np.min(o.t)
        try:
            col = {}
            c = 5
            for e in EPOCHS:                                             # :236-262
                fl = []
                for k in (1, 2, 3):
                    mu = gp.predict(p, np.array([[peak_time + e, WAVE[k]]]))[0] * scale
                    fl.append(mu)
                out[c:c + 3] = fl
                g, r, i = fl
                gr = -2.5 * np.log10(g / r) if (g > 0 and r > 0) else np.nan
                ri = -2.5 * np.log10(r / i) if (r > 0 and i > 0) else np.nan
                out[c + 3], out[c + 4] = gr, ri
                col[e] = gr
                c += 5
            if not np.isnan(col[0]) and not np.isnan(col[50]):           # :265-277
                out[25] = (col[50] - col[0]) / 50.0
            if not np.isnan(col[0]) and not np.isnan(col[100]):
                out[26] = (col[100] - col[0]) / 100.0
        except Exception:
            out[5:] = np.nan
    return out
