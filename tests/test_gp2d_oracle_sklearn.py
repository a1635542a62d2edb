"""Shrinks the unpinned surface of the 2-D GP oracle (oracle/gp2d.py, george absent).

The objective the oracle hands to L-BFGS-B -- the negative log marginal likelihood of a
constant-mean GP with kernel c * Matern-3/2(r), r^2 = dt^2/M0 + dl^2/M1, noise yerr^2 + 1.25e-12 --
and its analytic gradient are checked here against an INDEPENDENT implementation of the same
mathematics: scikit-learn's ``GaussianProcessRegressor`` with the anisotropic kernel
``ConstantKernel(c) * Matern(length_scale=[sqrt(M0), sqrt(M1)], nu=1.5)``, ``alpha = yerr^2 + 1.25e-12``,
target ``y - mu`` (multiband_gp.py:125-154).  Agreement to 1e-10 means the value and gradient
arithmetic of ``_GP.nll`` / ``_GP.grad_nll`` is right; what stays unverifiable offline is only
george's API convention, i.e. the three switches of oracle/gp2d.py:

* ``CONST_DIV_NDIM`` (``float * kernel`` -> ``ConstantKernel(log(float / ndim))``: the START point),
* ``PARAM_ORDER`` (which entries of the parameter vector the reference reads as its features),
* ``TINY`` (george's white-noise floor 1.25e-12).
"""
import numpy as np
import pytest

from oracle import gp2d
from oracle.common import iter_objects

sk = pytest.importorskip("sklearn.gaussian_process")


def _objects(golden_inputs, picks):
    objs = list(iter_objects(golden_inputs, golden_inputs["z"]))
    return [objs[i] for i in picks]


@pytest.mark.parametrize("pick", [0, 7, 33, 120, 210, 259])
def test_nll_and_gradient_match_sklearn(golden_inputs, pick):
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import ConstantKernel, Matern

    (o,) = _objects(golden_inputs, [pick])
    prep = gp2d.prepare(o)
    assert prep is not None
    X, y, yerr, _ = prep
    gp = gp2d._GP(X, y, yerr)
    rng = np.random.default_rng(pick)
    p0 = np.array([np.mean(y), np.log(np.var(y) / 2.0), np.log(100.0 ** 2), np.log(6000.0 ** 2)])
    for trial in range(4):
        p = p0 + (rng.normal(0, 0.7, 4) if trial else 0.0)
        mu, c, m0, m1 = p[0], np.exp(p[1]), np.exp(p[2]), np.exp(p[3])
        kern = ConstantKernel(c, constant_value_bounds="fixed") * Matern(
            length_scale=[np.sqrt(m0), np.sqrt(m1)], nu=1.5, length_scale_bounds="fixed")
        # free hyper-parameters for the gradient: theta = log(c, l_t, l_lambda)
        kern_free = ConstantKernel(c) * Matern(length_scale=[np.sqrt(m0), np.sqrt(m1)], nu=1.5)
        gpr = GaussianProcessRegressor(kernel=kern_free, alpha=yerr ** 2 + gp2d.TINY, optimizer=None,
                                       normalize_y=False).fit(X, y - mu)
        lml, grad = gpr.log_marginal_likelihood(gpr.kernel_.theta, eval_gradient=True)
        nll = gp.nll(p)
        g = gp.grad_nll(p)
        assert abs(nll + lml) <= 1e-10 * max(1.0, abs(lml)), (nll, lml)
        # d/d log M = 0.5 d/d log l ; objective = -LML
        want = np.array([-np.sum(gpr.alpha_), -grad[0], -0.5 * grad[1], -0.5 * grad[2]])
        assert np.allclose(g, want, rtol=1e-9, atol=1e-10 * max(1.0, np.abs(want).max())), (g, want)
        # and the prediction mean (multiband_gp.py:241-247): mu + k*' K^-1 (y - mu)
        xs = np.array([[X[:, 0].max() * 0.5, gp2d.WAVE[2]], [X[:, 0].max() + 20.0, gp2d.WAVE[1]]])
        fixed = GaussianProcessRegressor(kernel=kern, alpha=yerr ** 2 + gp2d.TINY, optimizer=None).fit(X, y - mu)
        assert np.allclose(gp.predict(p, xs), mu + fixed.predict(xs), rtol=1e-10, atol=1e-12)
