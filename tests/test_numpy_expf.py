"""numpy's float32 exp (the arithmetic of GridInterp's nearest-neighbour branch,
spec_inter.py:153-160) is not correctly rounded; csrc/common.h:np_expf restates
the algorithm numpy publishes for x86 hosts with AVX2+FMA / AVX-512
(loops_exponent_log.dispatch.c.src).  Here the same operation sequence, written
with numpy float32 arithmetic, is pinned to np.exp on this host bit for bit; the
GPU suite then pins the HIP function to np.exp."""
import numpy as np
import pytest

f32 = np.float32


def _fma(a, b, c):
    # float32 fma: the product of two float32 is exact in float64
    return (a.astype(np.float64) * b.astype(np.float64) +
            c.astype(np.float64)).astype(f32)


def np_expf_restated(x):
    x = x.astype(f32)
    c = lambda v: np.full_like(x, f32(v))   # noqa: E731
    q = (x * f32(1.44269504088896341)).astype(f32)
    q = ((q + f32(12582912.0)).astype(f32) - f32(12582912.0)).astype(f32)
    r = _fma(q, c(-6.93145752e-1), x)
    r = _fma(q, c(-1.42860677e-6), r)
    num = _fma(c(5.082762527590693718096e-04), r, c(6.757896990527504603057e-03))
    for k in (5.114512081637298353406e-02, 2.473615434895520810817e-01,
              7.257664613233124478488e-01, 9.999999999980870924916e-01):
        num = _fma(num, r, c(k))
    den = _fma(c(2.159509375685829852307e-02), r, c(-2.742335390411667452936e-01))
    den = _fma(den, r, c(1.0))
    return np.ldexp((num / den).astype(f32), q.astype(np.int32)).astype(f32)


def host_numpy_expf_is_published_algorithm():
    rng = np.random.RandomState(0)
    x = rng.uniform(-20, 20, 200000).astype(f32)
    return np.array_equal(np.exp(x), np_expf_restated(x))


def test_restated_expf_is_numpys():
    if not host_numpy_expf_is_published_algorithm():
        pytest.skip('this host\'s numpy float32 exp is not the AVX2 / AVX-512 '
                    'algorithm (no FMA unit?)')
    rng = np.random.RandomState(1)
    x = np.concatenate([rng.uniform(-87, 88, 1000000),
                        rng.normal(0, 2, 1000000)]).astype(f32)
    a, b = np.exp(x), np_expf_restated(x)
    assert np.array_equal(a, b)
    # and it is NOT the correctly rounded value in a large fraction of cases,
    # which is why a generic expf cannot stand in for it
    cr = np.exp(x.astype(np.float64)).astype(f32)
    assert 0.2 < np.mean(a != cr) < 0.6
