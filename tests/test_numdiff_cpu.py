"""The finite-difference Hessian of vel_fit.process (vel_fit.py:699-725).

numdifftools is a dependency of the reference that is absent from
/root/reference and from this image, so it is restated twice from its published
source: oracle/numdiff_restated.py (class for class, the checker) and
rvspecfit_amd/numdiff.py (batched, the product).  Pins available without the
package: the results its own docstrings publish, analytic Hessians, and the
structural facts the reference's two calls rest on.
"""
import numpy as np
import torch

from oracle import numdiff_restated as ref
from rvspecfit_amd import numdiff as nd


def rosen(x):
    return (1. - x[0])**2 + 105 * (x[1] - x[0]**2)**2


def rosen_hess(x):
    return np.array([[2 - 420 * (x[1] - x[0]**2) + 840 * x[0]**2, -420 * x[0]],
                     [-420 * x[0], 210.]])


def test_numdifftools_docstring_examples():
    """numdifftools.Hessian's docstring:
    >>> Hfun = nd.Hessian(rosen); Hfun([1, 1])
    array([[ 842., -420.], [-420.,  210.]])
    >>> nd.Hessian(lambda xy: np.cos(xy[0] - xy[1]))([0, 0])
    array([[-1.,  1.], [ 1., -1.]])"""
    h = ref.Hessian(rosen)([1, 1])
    np.testing.assert_allclose(h, [[842., -420.], [-420., 210.]], rtol=1e-9)
    h2 = ref.Hessian(lambda xy: np.cos(xy[0] - xy[1]))([0, 0])
    np.testing.assert_allclose(h2, [[-1., 1.], [1., -1.]], rtol=1e-9)


def test_min_step_generator_gives_one_exact_step():
    """MinStepGenerator(base_step=b) for a central Hessian (n = 2, order = 2):
    num_steps = max((n + order - 1) // 2, 1) + num_extrap = 1; the step is
    (b * max(log1p|x|, 1) + 1) - 1 -- no extrapolation in the first try"""
    b = np.array([0.01, 0.001, 0.001, 0.0001])
    x = np.array([5000., 2.5, -1.2, 0.3])
    g = ref.MinStepGenerator(base_step=b)
    steps = g(x, 'central', 2, 2)
    assert len(steps) == 1
    want = (b * np.maximum(np.log1p(np.abs(x)), 1.0) + 1.0) - 1.0
    np.testing.assert_array_equal(steps[0], want)
    got = nd.first_try_step(torch.as_tensor(b), torch.as_tensor(x)[None])
    np.testing.assert_array_equal(got[0].numpy(), want)
    # one step -> the plain central rule, exact on a quadratic form
    A = np.array([[3., 1, 0, 0], [1, 2, .5, 0], [0, .5, 1, .2], [0, 0, .2, 4]])
    x2 = np.array([1.5, 2.5, -1.2, 0.3])
    H = ref.Hessian(lambda p: 0.5 * p @ A @ p, step=g)(x2)
    np.testing.assert_allclose(H, A, rtol=1e-5, atol=1e-5)


def test_default_generator_is_15_shrinking_steps():
    x = np.array([5000., -1.])
    g = ref.Hessian(rosen).step
    steps = g(x, 'central', 2, 2)
    assert len(steps) == 15 and g.step_ratio == 1.6
    s0 = np.finfo(float).eps**(1 / 500.) * np.maximum(np.log1p(np.abs(x)), 1.)
    np.testing.assert_allclose(steps[0], s0, rtol=1e-15)
    np.testing.assert_allclose(steps[14], s0 / 1.6**14, rtol=1e-13)
    got = nd.retry_steps(torch.as_tensor(x)[None])[:, 0].numpy()
    np.testing.assert_allclose(got, np.array(steps), rtol=1e-14)


def test_product_follows_restatement_and_analytic_hessians():
    rng = np.random.RandomState(3)
    X = np.concatenate([[[1., 1.]], rng.uniform(-2, 2, size=(40, 2))])

    def f_b(idx, p):   # batched Rosenbrock (rows of p)
        return (1. - p[:, 0])**2 + 105 * (p[:, 1] - p[:, 0]**2)**2
    H = nd.hessian_retry(f_b, torch.as_tensor(X))
    for r in range(len(X)):
        Hr = ref.Hessian(rosen)(X[r])
        np.testing.assert_allclose(H[r], Hr, rtol=1e-12, atol=1e-12)
        # the function is a quartic: h^2 / h^4 Richardson terms remove the
        # truncation error altogether
        np.testing.assert_allclose(H[r], rosen_hess(X[r]), rtol=1e-8, atol=1e-7)

    # a non-polynomial function in 4 dimensions, first try and retry
    def g1(p):
        return np.exp(0.3 * p[0]) * np.sin(p[1]) + p[2]**2 * p[3] + \
            np.cos(p[0] * p[3])

    def g_b(idx, p):
        return torch.exp(0.3 * p[:, 0]) * torch.sin(p[:, 1]) + \
            p[:, 2]**2 * p[:, 3] + torch.cos(p[:, 0] * p[:, 3])
    X4 = rng.uniform(-1.5, 1.5, size=(12, 4))
    Hb = nd.hessian_retry(g_b, torch.as_tensor(X4))
    base = np.array([1e-3, 1e-3, 1e-3, 1e-3])
    h1 = nd.first_try_step(torch.as_tensor(base), torch.as_tensor(X4))
    H1 = nd.hessian_central(g_b, torch.as_tensor(X4), h1).numpy()
    for r in range(len(X4)):
        np.testing.assert_allclose(Hb[r], ref.Hessian(g1)(X4[r]), rtol=1e-9,
                                   atol=1e-10)
        np.testing.assert_allclose(
            H1[r], ref.Hessian(g1, step=ref.MinStepGenerator(base_step=base))(
                X4[r]), rtol=1e-7, atol=1e-8)
        # analytic check of the extrapolated result
        x = X4[r]
        s, c = np.sin(x[0] * x[3]), np.cos(x[0] * x[3])
        e = np.exp(0.3 * x[0])
        Ha = np.array([
            [0.09 * e * np.sin(x[1]) - x[3]**2 * c, 0.3 * e * np.cos(x[1]), 0,
             -s - x[0] * x[3] * c],
            [0.3 * e * np.cos(x[1]), -e * np.sin(x[1]), 0, 0],
            [0, 0, 2 * x[3], 2 * x[2]],
            [-s - x[0] * x[3] * c, 0, 2 * x[2], -x[0]**2 * c]])
        np.testing.assert_allclose(Hb[r], Ha, rtol=1e-7, atol=1e-8)


def test_noisy_function_picks_a_finite_estimate():
    """objective with 1e-9 of rounding noise (the regime of the real chi^2): the
    default-generator estimate stays close to the truth where one small step
    does not"""
    rng = np.random.RandomState(0)
    noise = {}

    def f(p):
        k = tuple(np.round(p, 14))
        if k not in noise:
            noise[k] = 1e-9 * rng.standard_normal()
        return 0.5 * (3 * p[0]**2 + p[0] * p[1] + 2 * p[1]**2) + noise[k]
    H = ref.Hessian(f)([0.3, -0.2])
    np.testing.assert_allclose(H, [[3., .5], [.5, 2.]], atol=1e-4)
