"""TEST INFRASTRUCTURE.  Restatement of the part of numdifftools (pbrod/numdifftools,
BSD-3, version 0.9.41 -- a dependency of the reference named in its
pyproject.toml but ABSENT from /root/reference and from this image) that
vel_fit.process uses (vel_fit.py:699-725):

    hess_step_gen = ndf.MinStepGenerator(base_step=hess_step)
    hessian = ndf.Hessian(hess_func_wrap, step=hess_step_gen)(x)       # 1st try
    hessian = ndf.Hessian(hess_func_wrap, step=None)(x)                # retry

Restated from the library's published source (numdifftools/step_generators.py,
core.py, limits.py, extrapolation.py); it cannot be run here, so this file is
"parity unpinned" against the real package.  Known-answer checks: the results
the library's own docstrings publish (Rosenbrock Hessian at (1,1), cos(x-y) at
the origin) in tests/test_numdiff_cpu.py.

What the two calls amount to:
* MinStepGenerator(base_step=b): num_steps defaults to min_num_steps + num_extrap
  = max((n + order - 1) // 2, 1) + 0 = 1 for the Hessian (n = 2, order = 2,
  'central'): ONE step h = make_exact(b * nominal_step(x)),
  nominal_step = log1p(|x|).clip(min=1).  With one step the Richardson rule is
  [1], Wynn's epsilon step needs three estimates and is skipped, and the result
  is the plain central formula (eq. 9 of the Hessian docstring).
* step=None, method 'central': MaxStepGenerator(base_step=None, step_ratio=None,
  num_extrap=0): 15 steps base * 1.6**(-i), base = EPS**(1/500) * nominal_step(x)
  (scale 500, use_exact_steps False); the 15 Hessian estimates go through
  Richardson extrapolation (3-term rule for error terms h^2, h^4), one Wynn
  epsilon (dea3) pass, and the estimate of smallest error is picked per matrix
  element after outlier trimming.
"""
import warnings

import numpy as np
from scipy import linalg
from scipy.ndimage import convolve1d

EPS = np.finfo(float).eps
TINY = np.finfo(float).tiny


def make_exact(h):
    """h + 1 - 1: a step that is exactly representable next to O(1) numbers"""
    return (h + 1.0) - 1.0


def nominal_step(x=None):
    if x is None:
        return 1.0
    return np.log1p(np.abs(x)).clip(min=1.0)


def default_scale(method='forward', n=1, order=2):
    high_order = int(n > 1 or order >= 4)
    order2 = max(order // 2 - 1, 0)
    n4 = n // 4
    n_mod_4 = n % 4
    c = ([n4 * (10 + 1.5 * int(n > 10)),
          3.65 + n4 * (5 + 1.5**n4),
          3.65 + n4 * (5 + 1.7**n4),
          7.30 + n4 * (5 + 2.1**n4)][n_mod_4]) if high_order else 0
    return (dict(multicomplex=1.06, complex=1.06 + c).get(method, 2.5) +
            int(n - 1) * dict(multicomplex=0, complex=0.0).get(method, 1.3) +
            order2 * dict(central=3, forward=2, backward=2).get(method, 0))


class MinStepGenerator:
    """steps = step_nom * base_step * step_ratio**(i + offset),
    i = num_steps-1, ..., 1, 0"""
    _sign = 1

    def __init__(self, base_step=None, step_ratio=None, num_steps=None,
                 step_nom=None, offset=0, num_extrap=0, use_exact_steps=True,
                 check_num_steps=True, scale=None):
        self._base_step = base_step
        self._step_ratio = step_ratio
        self._num_steps = num_steps
        self._step_nom = step_nom
        self.offset = offset
        self.num_extrap = num_extrap
        self.use_exact_steps = use_exact_steps
        self.check_num_steps = check_num_steps
        self._scale = scale
        self._state = (np.asarray(1), 'forward', 1, 2)

    @property
    def scale(self):
        if self._scale is None:
            _x, method, n, order = self._state
            return default_scale(method, n, order)
        return self._scale

    @property
    def base_step(self):
        if self._base_step is None:
            return EPS**(1. / self.scale)
        return np.asarray(self._base_step, dtype=float)

    @property
    def step_nom(self):
        x = self._state[0]
        if self._step_nom is None:
            return nominal_step(x)
        return np.full(np.shape(x), self._step_nom, dtype=float)

    @property
    def step_ratio(self):
        r = self._step_ratio
        if r is None:
            r = {1: 2.0}.get(self._state[2], 1.6)
        return float(r)

    @property
    def min_num_steps(self):
        _x, method, n, order = self._state
        num_steps = int(n + order - 1)
        if method in ('central', 'central2', 'complex', 'multicomplex'):
            step = 2
            if method == 'complex':
                step = 4 if (n > 2 or order >= 4) else 2
            num_steps = (n + order - 1) // step
        return max(int(num_steps), 1)

    @property
    def num_steps(self):
        mn = self.min_num_steps
        if self._num_steps is not None:
            ns = int(self._num_steps)
            if self.check_num_steps:
                ns = max(ns, mn)
            return ns
        return mn + int(self.num_extrap)

    def _range(self):
        return range(self.num_steps - 1, -1, -1)

    def __call__(self, x=None, method='central', n=1, order=2):
        self._state = (np.asarray(x, dtype=float), method, n, order)
        base_step, step_ratio = self.base_step * self.step_nom, self.step_ratio
        if self.use_exact_steps:
            base_step, step_ratio = make_exact(base_step), make_exact(step_ratio)
        out = []
        for i in self._range():
            step = base_step * step_ratio**(self._sign * i + self.offset)
            if (np.abs(step) > 0).all():
                out.append(step)
        return out


class MaxStepGenerator(MinStepGenerator):
    """steps = step_nom * base_step * step_ratio**(-i + offset),
    i = 0, 1, ..., num_steps-1"""
    _sign = -1

    def __init__(self, base_step=2.0, step_ratio=2.0, num_steps=15,
                 step_nom=None, offset=0, num_extrap=0, use_exact_steps=False,
                 check_num_steps=True, scale=500):
        super().__init__(base_step=base_step, step_ratio=step_ratio,
                         num_steps=num_steps, step_nom=step_nom, offset=offset,
                         num_extrap=num_extrap, use_exact_steps=use_exact_steps,
                         check_num_steps=check_num_steps, scale=scale)

    def _range(self):
        return range(self.num_steps)


def max_abs(a, b):
    return np.maximum(np.abs(a), np.abs(b))


def dea3(v0, v1, v2, symmetric=False):
    """one step of Wynn's epsilon algorithm on three consecutive estimates"""
    e0, e1, e2 = np.atleast_1d(v0, v1, v2)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        with np.errstate(all='ignore'):
            delta2, delta1 = e2 - e1, e1 - e0
            err2, err1 = abs(delta2), abs(delta1)
            tol2, tol1 = max_abs(e2, e1) * EPS, max_abs(e1, e0) * EPS
            delta1[err1 < TINY] = TINY
            delta2[err2 < TINY] = TINY
            ss = 1.0 / delta2 - 1.0 / delta1 + TINY
            smalle2 = abs(ss * e1) <= 1.0e-3
            converged = (err1 <= tol1) & (err2 <= tol2) | smalle2
            result = np.where(converged, e2 * 1.0, e1 + 1.0 / ss)
    abserr = err1 + err2 + np.where(converged, tol2 * 10, abs(result - e2))
    if symmetric and len(result) > 1:
        return result[:-1], abserr[1:]
    return result, abserr


class Richardson:

    def __init__(self, step_ratio=2.0, step=1, order=1, num_terms=2):
        self.num_terms = num_terms
        self.order = order
        self.step = step
        self.step_ratio = step_ratio

    def _r_matrix(self, num_terms):
        step = self.step
        i, j = np.ogrid[0:num_terms + 1, 0:num_terms]
        r_mat = np.ones((num_terms + 1, num_terms + 1))
        r_mat[:, 1:] = (1.0 / self.step_ratio)**(i * (step * j + self.order))
        return r_mat

    def rule(self, sequence_length=None):
        if sequence_length is None:
            sequence_length = self.num_terms + 1
        num_terms = min(self.num_terms, sequence_length - 1)
        if num_terms > 0:
            return linalg.pinv(self._r_matrix(num_terms))[0]
        return np.ones((1, ))

    @staticmethod
    def _estimate_error(new_sequence, old_sequence, steps, rule):
        m = new_sequence.shape[0]
        mo = old_sequence.shape[0]
        cov1 = np.sum(rule**2)
        fact = np.maximum(12.7062047361747 * np.sqrt(cov1), EPS * 10.)
        if mo < 2:
            return (np.abs(new_sequence) * EPS + steps) * fact
        if m < 2:
            delta = np.diff(old_sequence, axis=0)
            tol = max_abs(old_sequence[:-1], old_sequence[1:]) * fact
            err = np.abs(delta)
            converged = err <= tol
            return err[-m:] + np.where(converged[-m:], tol[-m:] * 10,
                                       abs(new_sequence -
                                           old_sequence[-m:]) * fact)
        err = np.abs(np.diff(new_sequence, axis=0)) * fact
        tol = max_abs(new_sequence[1:], new_sequence[:-1]) * EPS * fact
        converged = err <= tol
        return err + np.where(converged, tol * 10,
                              abs(new_sequence[:-1] - old_sequence[1:m]) * fact)

    def __call__(self, sequence, steps):
        ne = sequence.shape[0]
        rule = self.rule(ne)
        nr = rule.size - 1
        m = ne - nr
        mm = min(ne, m + 1)
        new_sequence = convolve1d(np.asarray(sequence, dtype=float), rule[::-1],
                                  axis=0, origin=nr // 2)
        abserr = self._estimate_error(new_sequence[:mm], sequence, steps, rule)
        return new_sequence[:m], abserr[:m], steps[:m]


def _add_error_to_outliers(der, trim_fact=10):
    try:
        median = np.nanmedian(der, axis=0)
        p75 = np.nanpercentile(der, 75, axis=0)
        p25 = np.nanpercentile(der, 25, axis=0)
        iqr = np.abs(p75 - p25)
    except ValueError as msg:
        warnings.warn(str(msg))
        return 0 * der
    a_median = np.abs(median)
    outliers = (((abs(der) < (a_median / trim_fact)) +
                 (abs(der) > (a_median * trim_fact))) * (a_median > 1e-8) +
                ((der < p25 - 1.5 * iqr) + (p75 + 1.5 * iqr < der)))
    return outliers * np.abs(der - median)


def _get_arg_min(errors):
    shape = errors.shape
    try:
        arg_mins = np.nanargmin(errors, axis=0)
        min_errors = np.nanmin(errors, axis=0)
    except ValueError as msg:
        warnings.warn(str(msg))
        return np.arange(shape[1])
    for i, min_error in enumerate(min_errors):
        idx = np.flatnonzero(errors[:, i] == min_error)
        arg_mins[i] = idx[idx.size // 2]
    return np.ravel_multi_index((arg_mins, np.arange(shape[1])), shape)


class Hessian:
    """numdifftools.Hessian(f, step=None, method='central')"""

    def __init__(self, f, step=None, method='central'):
        assert method == 'central'
        self.f = f
        self.method = method
        self.n = 2
        self.order = 2
        self.richardson_terms = 2
        if hasattr(step, '__call__'):
            self.step = step
        else:
            step_nom = None if step is None else 1
            self.step = MaxStepGenerator(base_step=step, step_ratio=None,
                                         num_extrap=0, step_nom=step_nom)

    @staticmethod
    def _central_even(f, f_x0i, x0i, h):
        n = len(x0i)
        ee = np.diag(h)
        hess = np.empty((n, n), dtype=float)
        np.outer(h, h, out=hess)
        for i in range(n):
            hess[i, i] = (f(x0i + 2 * ee[i, :]) - 2 * f_x0i +
                          f(x0i - 2 * ee[i, :])) / (4. * hess[i, i])
            for j in range(i + 1, n):
                hess[i, j] = (f(x0i + ee[i, :] + ee[j, :]) -
                              f(x0i + ee[i, :] - ee[j, :]) -
                              f(x0i - ee[i, :] + ee[j, :]) +
                              f(x0i - ee[i, :] - ee[j, :])) / (4. * hess[j, i])
                hess[j, i] = hess[i, j]
        return hess

    def __call__(self, x):
        xi = np.asarray(x, dtype=float)
        steps = self.step(xi, self.method, self.n, self.order)
        step_ratio = self.step.step_ratio
        fxi = self.f(xi)
        results = [self._central_even(self.f, fxi, xi, h) for h in steps]
        shape = list(results[0].shape)
        f_del = np.vstack([r.ravel() for r in results])
        one = np.ones(shape)
        hh = np.vstack([(one * h).ravel() for h in steps])
        rich = Richardson(step_ratio=step_ratio, step=2, order=2,
                          num_terms=self.richardson_terms)
        der1, err1, hh = rich(f_del, hh)
        if len(der1) > 2:
            der1, err1 = dea3(der1[0:-2], der1[1:-1], der1[2:], symmetric=False)
            hh = hh[2:]
        err1 = err1 + _add_error_to_outliers(der1)
        ix = _get_arg_min(err1)
        return der1.flat[ix].reshape(shape)
