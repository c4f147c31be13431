"""Lock-step BFGS: the optional `second_minimizer` polish of vel_fit.process
(vel_fit.py:653-658: scipy.optimize.minimize(chisq_func, x_NM, method='BFGS',
options=dict(hess_inv0=...)) with a forward-difference gradient).

scipy's algorithm is restated as ONE Python generator per spectrum -- the BFGS
iteration of scipy/optimize/_optimize.py::_minimize_bfgs, the Wolfe line search
of _linesearch.py (line_search_wolfe1 = MINPACK-2 dcsrch/dcstep, falling back to
line_search_wolfe2 with its cubic/quadratic zoom) and the 2-point finite
differences of _numdiff.approx_derivative (absolute step sqrt(eps)).  A
generator yields the points it needs the objective at and is sent the values;
the driver collects the requests of all spectra and evaluates them in batched
launches.  Control flow and the order of every floating-point operation follow
scipy (1.15), so on equal function values the iterates are identical
(tests/test_tools_cpu.py compares with scipy itself).

The per-spectrum state (x, gradient, inverse Hessian: n <= 8) lives on the host,
as in the reference; only the objective runs on the GPU.
"""
import math

import numpy as np

_EPS = float(np.sqrt(np.finfo(float).eps))   # scipy's _epsilon


class _Fail(Exception):
    """scipy's _LineSearchError"""


# --------------------------------------------------------------------------
# objective with scipy's ScalarFunction caching: f and g of the latest x
# --------------------------------------------------------------------------
class _SF:

    def __init__(self):
        self.x = None
        self.f = None
        self.g = None
        self.nfev = 0
        self.ngev = 0

    def _set_x(self, x):
        if self.x is None or not (x == self.x).all():
            self.x = np.array(x, dtype=float, copy=True)
            self.f = None
            self.g = None

    def fun(self, x):
        self._set_x(x)
        if self.f is None:
            self.f = yield ('f', self.x)
            self.nfev += 1
        return self.f

    def _fd_points(self):
        # _numdiff.approx_derivative(method='2-point', abs_step=_EPS)
        x0 = self.x
        n = len(x0)
        dx = (x0 + _EPS) - x0
        if (dx == 0).any():   # |x| > 1e8: fall back to a relative step
            sign_x0 = (x0 >= 0).astype(float) * 2 - 1
            h = np.where(dx == 0,
                         _EPS * sign_x0 * np.maximum(1.0, np.abs(x0)), _EPS)
            return x0[None, :] + np.diag(h)
        return x0[None, :] + _diag_eps(n)

    def grad(self, x):
        self._set_x(x)
        if self.g is None:
            if self.f is None:
                self.f = yield ('f', self.x)
                self.nfev += 1
            x0 = self.x
            n = len(x0)
            x1 = self._fd_points()
            f1 = yield ('g', x1)
            self.nfev += n
            self.ngev += 1
            dxi = x1.ravel()[::n + 1] - x0
            self.g = (np.asarray(f1) - self.f) / dxi
        return self.g

    def fun_grad(self, x):
        """fun(x) followed by grad(x) as scipy calls them, but as ONE request
        of 1 + n points when neither is cached: the same points, the same
        values, the same counters -- half the rounds of the lock-step driver"""
        self._set_x(x)
        if self.f is None and self.g is None:
            x0 = self.x
            n = len(x0)
            x1 = self._fd_points()
            vals = yield ('fg', np.concatenate([x0[None, :], x1], axis=0))
            self.f = float(vals[0])
            self.nfev += 1 + n
            self.ngev += 1
            dxi = x1.ravel()[::n + 1] - x0
            self.g = (np.asarray(vals[1:]) - self.f) / dxi
            return self.f, self.g
        f = yield from self.fun(x)
        g = yield from self.grad(x)
        return f, g


_DIAG = {}


def _diag_eps(n):
    if n not in _DIAG:
        _DIAG[n] = np.diag(np.full(n, _EPS))
    return _DIAG[n]


# --------------------------------------------------------------------------
# MINPACK-2 dcstep / dcsrch (More' & Thuente) as scipy's _dcsrch.py states them
# --------------------------------------------------------------------------
def _sign(x):
    return (x > 0) - (x < 0) if x == x else x


def _sqrt(x):
    # np.sqrt of a negative / nan argument is nan (warnings silenced in scipy)
    return math.sqrt(x) if x >= 0 else float('nan')


def _clip(x, lo, hi):
    return min(max(x, lo), hi)   # nan stays nan, like np.clip


def _dcstep(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, stpmin, stpmax):
    """scalar Python-float arithmetic (the same IEEE operations as scipy's numpy
    scalars, ~10x cheaper); zero divisions give inf/nan like numpy's"""
    stx, fx, dx, sty, fy, dy, stp, fp, dp = (
        float(stx), float(fx), float(dx), float(sty), float(fy), float(dy),
        float(stp), float(fp), float(dp))
    sgnd = _sign(dp) * _sign(dx)
    try:
        return _dcstep_core(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt,
                            stpmin, stpmax, sgnd)
    except (ZeroDivisionError, OverflowError):
        with np.errstate(all='ignore'):
            f64 = np.float64
            return _dcstep_core(f64(stx), f64(fx), f64(dx), f64(sty), f64(fy),
                                f64(dy), f64(stp), f64(fp), f64(dp), brackt,
                                stpmin, stpmax, sgnd, sqrt=np.sqrt)


def _dcstep_core(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, stpmin, stpmax,
                 sgnd, sqrt=_sqrt):
    if True:
        if fp > fx:
            theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp
            s = max(abs(theta), abs(dx), abs(dp))
            gamma = s * sqrt((theta / s)**2 - (dx / s) * (dp / s))
            if stp < stx:
                gamma *= -1
            p = (gamma - dx) + theta
            q = ((gamma - dx) + gamma) + dp
            r = p / q
            stpc = stx + r * (stp - stx)
            stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * \
                (stp - stx)
            if abs(stpc - stx) <= abs(stpq - stx):
                stpf = stpc
            else:
                stpf = stpc + (stpq - stpc) / 2.0
            brackt = True
        elif sgnd < 0.0:
            theta = 3 * (fx - fp) / (stp - stx) + dx + dp
            s = max(abs(theta), abs(dx), abs(dp))
            gamma = s * sqrt((theta / s)**2 - (dx / s) * (dp / s))
            if stp > stx:
                gamma *= -1
            p = (gamma - dp) + theta
            q = ((gamma - dp) + gamma) + dx
            r = p / q
            stpc = stp + r * (stx - stp)
            stpq = stp + (dp / (dp - dx)) * (stx - stp)
            if abs(stpc - stp) > abs(stpq - stp):
                stpf = stpc
            else:
                stpf = stpq
            brackt = True
        elif abs(dp) < abs(dx):
            theta = 3 * (fx - fp) / (stp - stx) + dx + dp
            s = max(abs(theta), abs(dx), abs(dp))
            gamma = s * sqrt(max(0, (theta / s)**2 - (dx / s) * (dp / s)))
            if stp > stx:
                gamma = -gamma
            p = (gamma - dp) + theta
            q = (gamma + (dx - dp)) + gamma
            r = p / q
            if r < 0 and gamma != 0:
                stpc = stp + r * (stx - stp)
            elif stp > stx:
                stpc = stpmax
            else:
                stpc = stpmin
            stpq = stp + (dp / (dp - dx)) * (stx - stp)
            if brackt:
                if abs(stpc - stp) < abs(stpq - stp):
                    stpf = stpc
                else:
                    stpf = stpq
                if stp > stx:
                    stpf = min(stp + 0.66 * (sty - stp), stpf)
                else:
                    stpf = max(stp + 0.66 * (sty - stp), stpf)
            else:
                if abs(stpc - stp) > abs(stpq - stp):
                    stpf = stpc
                else:
                    stpf = stpq
                stpf = _clip(stpf, stpmin, stpmax)
        else:
            if brackt:
                theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp
                s = max(abs(theta), abs(dy), abs(dp))
                gamma = s * sqrt((theta / s)**2 - (dy / s) * (dp / s))
                if stp > sty:
                    gamma = -gamma
                p = (gamma - dp) + theta
                q = ((gamma - dp) + gamma) + dy
                r = p / q
                stpc = stp + r * (sty - stp)
                stpf = stpc
            elif stp > stx:
                stpf = stpmax
            else:
                stpf = stpmin
    if fp > fx:
        sty, fy, dy = stp, fp, dp
    else:
        if sgnd < 0:
            sty, fy, dy = stx, fx, dx
        stx, fx, dx = stp, fp, dp
    return stx, fx, dx, sty, fy, dy, stpf, brackt


class _Dcsrch:
    """one More'-Thuente search: step(stp, f, g) -> (stp, task)"""

    def __init__(self, ftol, gtol, xtol, stpmin, stpmax):
        self.ftol, self.gtol, self.xtol = ftol, gtol, xtol
        self.stpmin, self.stpmax = stpmin, stpmax
        self.started = False

    def step(self, stp, f, g):
        p5, p66, xtrapl, xtrapu = 0.5, 0.66, 1.1, 4.0
        if not self.started:
            self.started = True
            task = 'FG'
            if stp < self.stpmin:
                task = 'ERROR'
            if stp > self.stpmax:
                task = 'ERROR'
            if g >= 0:
                task = 'ERROR'
            if task == 'ERROR':
                return stp, task
            self.brackt = False
            self.stage = 1
            self.finit, self.ginit = f, g
            self.gtest = self.ftol * self.ginit
            self.width = self.stpmax - self.stpmin
            self.width1 = self.width / p5
            self.stx, self.fx, self.gx = 0.0, self.finit, self.ginit
            self.sty, self.fy, self.gy = 0.0, self.finit, self.ginit
            self.stmin = 0
            self.stmax = stp + xtrapu * stp
            return stp, 'FG'
        task = 'FG'
        ftest = self.finit + stp * self.gtest
        if self.stage == 1 and f <= ftest and g >= 0:
            self.stage = 2
        if self.brackt and (stp <= self.stmin or stp >= self.stmax):
            task = 'WARN'
        if self.brackt and self.stmax - self.stmin <= self.xtol * self.stmax:
            task = 'WARN'
        if stp == self.stpmax and f <= ftest and g <= self.gtest:
            task = 'WARN'
        if stp == self.stpmin and (f > ftest or g >= self.gtest):
            task = 'WARN'
        if f <= ftest and abs(g) <= self.gtol * -self.ginit:
            task = 'CONV'
        if task != 'FG':
            return stp, task
        if self.stage == 1 and f <= self.fx and f > ftest:
            fm = f - stp * self.gtest
            fxm = self.fx - self.stx * self.gtest
            fym = self.fy - self.sty * self.gtest
            gm = g - self.gtest
            gxm = self.gx - self.gtest
            gym = self.gy - self.gtest
            (self.stx, fxm, gxm, self.sty, fym, gym, stp,
             self.brackt) = _dcstep(self.stx, fxm, gxm, self.sty, fym, gym, stp,
                                    fm, gm, self.brackt, self.stmin, self.stmax)
            self.fx = fxm + self.stx * self.gtest
            self.fy = fym + self.sty * self.gtest
            self.gx = gxm + self.gtest
            self.gy = gym + self.gtest
        else:
            (self.stx, self.fx, self.gx, self.sty, self.fy, self.gy, stp,
             self.brackt) = _dcstep(self.stx, self.fx, self.gx, self.sty,
                                    self.fy, self.gy, stp, f, g, self.brackt,
                                    self.stmin, self.stmax)
        if self.brackt:
            if abs(self.sty - self.stx) >= p66 * self.width1:
                stp = self.stx + p5 * (self.sty - self.stx)
            self.width1 = self.width
            self.width = abs(self.sty - self.stx)
        if self.brackt:
            self.stmin = min(self.stx, self.sty)
            self.stmax = max(self.stx, self.sty)
        else:
            self.stmin = stp + xtrapl * (stp - self.stx)
            self.stmax = stp + xtrapu * (stp - self.stx)
        stp = _clip(stp, self.stpmin, self.stpmax)
        if (self.brackt and (stp <= self.stmin or stp >= self.stmax)
                or (self.brackt
                    and self.stmax - self.stmin <= self.xtol * self.stmax)):
            stp = self.stx
        return stp, 'FG'


def _wolfe1(sf, xk, pk, gfk, old_fval, old_old_fval, c1, c2, amax, amin,
            xtol=1e-14):
    """line_search_wolfe1 / scalar_search_wolfe1; returns
    (stp | None, fval, old_fval, gval)"""
    gval = gfk
    derphi0 = np.dot(gfk, pk)
    phi0 = old_fval
    if old_old_fval is not None and derphi0 != 0:
        alpha1 = min(1.0, 1.01 * 2 * (phi0 - old_old_fval) / derphi0)
        if alpha1 < 0:
            alpha1 = 1.0
    else:
        alpha1 = 1.0
    ds = _Dcsrch(c1, c2, xtol, amin, amax)
    phi1, derphi1 = phi0, derphi0
    stp = None
    task = 'START'
    for _ in range(100):
        stp, task = ds.step(alpha1, phi1, derphi1)
        if not math.isfinite(stp):
            task = 'WARN'
            stp = None
            break
        if task == 'FG':
            alpha1 = stp
            phi1, gval = yield from sf.fun_grad(xk + stp * pk)
            derphi1 = np.dot(gval, pk)
        else:
            break
    else:
        stp = None
        task = 'WARN'
    if task in ('ERROR', 'WARN'):
        stp = None
    return stp, phi1, phi0, gval


def _cubicmin(a, fa, fpa, b, fb, c, fc):
    with np.errstate(divide='raise', over='raise', invalid='raise'):
        try:
            C = fpa
            db = b - a
            dc = c - a
            denom = (db * dc)**2 * (db - dc)
            d1 = np.empty((2, 2))
            d1[0, 0] = dc**2
            d1[0, 1] = -db**2
            d1[1, 0] = -dc**3
            d1[1, 1] = db**3
            [A, B] = np.dot(d1, np.asarray([fb - fa - C * db,
                                            fc - fa - C * dc]).flatten())
            A /= denom
            B /= denom
            radical = B * B - 3 * A * C
            xmin = a + (-B + np.sqrt(radical)) / (3 * A)
        except ArithmeticError:
            return None
    if not np.isfinite(xmin):
        return None
    return xmin


def _quadmin(a, fa, fpa, b, fb):
    with np.errstate(divide='raise', over='raise', invalid='raise'):
        try:
            D = fa
            C = fpa
            db = b - a * 1.0
            B = (fb - D - C * db) / (db * db)
            xmin = a - C / (2.0 * B)
        except ArithmeticError:
            return None
    if not np.isfinite(xmin):
        return None
    return xmin


def _wolfe2(sf, xk, pk, gfk, old_fval, old_old_fval, c1, c2, amax, maxiter=10):
    """line_search_wolfe2 / scalar_search_wolfe2 / _zoom; returns
    (alpha | None, phi_star, old_fval, gval | None)"""
    gval = [None]

    def phi(alpha):
        return (yield from sf.fun(xk + alpha * pk))

    def derphi(alpha):
        gval[0] = yield from sf.grad(xk + alpha * pk)
        return np.dot(gval[0], pk)

    derphi0 = np.dot(gfk, pk)
    phi0, old_phi0 = old_fval, old_old_fval
    alpha0 = 0
    if old_phi0 is not None and derphi0 != 0:
        alpha1 = min(1.0, 1.01 * 2 * (phi0 - old_phi0) / derphi0)
    else:
        alpha1 = 1.0
    if alpha1 < 0:
        alpha1 = 1.0
    if amax is not None:
        alpha1 = min(alpha1, amax)
    phi_a1 = yield from phi(alpha1)
    phi_a0 = phi0
    derphi_a0 = derphi0

    def zoom(a_lo, a_hi, phi_lo, phi_hi, derphi_lo):
        i = 0
        delta1, delta2 = 0.2, 0.1
        phi_rec, a_rec = phi0, 0
        a_j = None
        while True:
            dalpha = a_hi - a_lo
            if dalpha < 0:
                a, b = a_hi, a_lo
            else:
                a, b = a_lo, a_hi
            if i > 0:
                cchk = delta1 * dalpha
                a_j = _cubicmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi, a_rec,
                                phi_rec)
            if (i == 0) or (a_j is None) or (a_j > b - cchk) or \
                    (a_j < a + cchk):
                qchk = delta2 * dalpha
                a_j = _quadmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi)
                if (a_j is None) or (a_j > b - qchk) or (a_j < a + qchk):
                    a_j = a_lo + 0.5 * dalpha
            phi_aj = yield from phi(a_j)
            if (phi_aj > phi0 + c1 * a_j * derphi0) or (phi_aj >= phi_lo):
                phi_rec, a_rec = phi_hi, a_hi
                a_hi, phi_hi = a_j, phi_aj
            else:
                derphi_aj = yield from derphi(a_j)
                if abs(derphi_aj) <= -c2 * derphi0:
                    return a_j, phi_aj, derphi_aj
                if derphi_aj * (a_hi - a_lo) >= 0:
                    phi_rec, a_rec = phi_hi, a_hi
                    a_hi, phi_hi = a_lo, phi_lo
                else:
                    phi_rec, a_rec = phi_lo, a_lo
                a_lo, phi_lo, derphi_lo = a_j, phi_aj, derphi_aj
            i += 1
            if i > 10:
                return None, None, None

    alpha_star = phi_star = derphi_star = None
    for i in range(maxiter):
        if alpha1 == 0 or (amax is not None and alpha0 > amax):
            alpha_star, phi_star, derphi_star = None, phi0, None
            phi0 = old_phi0
            break
        if (phi_a1 > phi0 + c1 * alpha1 * derphi0) or \
                ((phi_a1 >= phi_a0) and i > 0):
            alpha_star, phi_star, derphi_star = yield from zoom(
                alpha0, alpha1, phi_a0, phi_a1, derphi_a0)
            break
        derphi_a1 = yield from derphi(alpha1)
        if abs(derphi_a1) <= -c2 * derphi0:
            alpha_star, phi_star, derphi_star = alpha1, phi_a1, derphi_a1
            break
        if derphi_a1 >= 0:
            alpha_star, phi_star, derphi_star = yield from zoom(
                alpha1, alpha0, phi_a1, phi_a0, derphi_a1)
            break
        alpha2 = 2 * alpha1
        if amax is not None:
            alpha2 = min(alpha2, amax)
        alpha0, alpha1 = alpha1, alpha2
        phi_a0 = phi_a1
        phi_a1 = yield from phi(alpha1)
        derphi_a0 = derphi_a1
    else:
        alpha_star, phi_star, derphi_star = alpha1, phi_a1, None
    return alpha_star, phi_star, phi0, \
        (None if derphi_star is None else gval[0])


def bfgs_generator(x0, hess_inv0=None, gtol=1e-5, c1=1e-4, c2=0.9, xrtol=0,
                   maxiter=None):
    """_minimize_bfgs for one starting point as a request generator; its return
    value (StopIteration.value) is dict(x, fun, jac, hess_inv, nit, nfev, njev,
    status, success)."""
    sf = _SF()
    x0 = np.asarray(x0, dtype=float).flatten()
    N = len(x0)
    if maxiter is None:
        maxiter = N * 200
    old_fval, gfk = yield from sf.fun_grad(x0)
    k = 0
    I = np.eye(N, dtype=int)
    Hk = I if hess_inv0 is None else hess_inv0
    old_old_fval = old_fval + np.linalg.norm(gfk) / 2
    xk = x0
    warnflag = 0
    gnorm = np.amax(np.abs(gfk))
    while (gnorm > gtol) and (k < maxiter):
        pk = -np.dot(Hk, gfk)
        try:
            # _line_search_wolfe12(amin=1e-100, amax=1e100)
            stp, fval, ofv, gfkp1 = yield from _wolfe1(
                sf, xk, pk, gfk, old_fval, old_old_fval, c1, c2, 1e100, 1e-100)
            if stp is None:
                stp, fval, ofv, gfkp1 = yield from _wolfe2(
                    sf, xk, pk, gfk, old_fval, old_old_fval, c1, c2, 1e100)
            if stp is None:
                raise _Fail()
            alpha_k = stp
            old_fval, old_old_fval = fval, ofv
        except _Fail:
            warnflag = 2
            break
        sk = alpha_k * pk
        xkp1 = xk + sk
        xk = xkp1
        if gfkp1 is None:
            gfkp1 = yield from sf.grad(xkp1)
        yk = gfkp1 - gfk
        gfk = gfkp1
        k += 1
        gnorm = np.amax(np.abs(gfk))
        if gnorm <= gtol:
            break
        if alpha_k * np.sqrt(np.sum(pk**2)) <= xrtol * (
                xrtol + np.sqrt(np.sum(xk**2))):
            break
        if not np.isfinite(old_fval):
            warnflag = 2
            break
        rhok_inv = np.dot(yk, sk)
        if rhok_inv == 0.:
            rhok = 1000.0
        else:
            rhok = 1. / rhok_inv
        A1 = I - sk[:, np.newaxis] * yk[np.newaxis, :] * rhok
        A2 = I - yk[:, np.newaxis] * sk[np.newaxis, :] * rhok
        Hk = np.dot(A1, np.dot(Hk, A2)) + (rhok * sk[:, np.newaxis] *
                                            sk[np.newaxis, :])
    fval = old_fval
    if warnflag == 2:
        pass
    elif k >= maxiter:
        warnflag = 1
    elif np.isnan(gnorm) or np.isnan(fval) or np.isnan(xk).any():
        warnflag = 3
    return dict(x=xk, fun=fval, jac=gfk, hess_inv=Hk, nit=k, nfev=sf.nfev,
                njev=sf.ngev, status=warnflag, success=(warnflag == 0))


def minimize_lockstep(func, x0, hess_inv0=None, max_rows=None, **kw):
    """BFGS from every row of x0 [S, n] (numpy).  func(idx int64 [J], X [J, n])
    -> f [J] (numpy in, numpy out) is called with the requests of all spectra
    that are waiting, at most max_rows rows at a time.
    Returns dict(x [S,n], fun [S], nit [S], nfev [S], status [S])."""
    x0 = np.asarray(x0, dtype=float)
    S, n = x0.shape
    gens = [bfgs_generator(x0[i], hess_inv0=hess_inv0, **kw) for i in range(S)]
    pending = {}
    results = [None] * S
    for i, g in enumerate(gens):
        try:
            pending[i] = next(g)
        except StopIteration as e:   # cannot happen: the first request is f(x0)
            results[i] = e.value
    rounds = 0
    while pending:
        rounds += 1
        ids = sorted(pending)
        rows_idx, rows_x, spans = [], [], []
        for i in ids:
            kind, x = pending[i]
            x = np.atleast_2d(x)
            spans.append((len(rows_idx), len(rows_idx) + len(x), kind))
            rows_idx.extend([i] * len(x))
            rows_x.append(x)
        X = np.concatenate(rows_x, axis=0)
        idx = np.asarray(rows_idx, dtype=np.int64)
        if max_rows is None or len(idx) <= max_rows:
            F = np.asarray(func(idx, X), dtype=float)
        else:
            F = np.concatenate([
                np.asarray(func(idx[a:a + max_rows], X[a:a + max_rows]),
                           dtype=float)
                for a in range(0, len(idx), max_rows)])
        for i, (a, b, kind) in zip(ids, spans):
            val = float(F[a]) if kind == 'f' else F[a:b].copy()
            try:
                pending[i] = gens[i].send(val)
            except StopIteration as e:
                results[i] = e.value
                del pending[i]
    return dict(x=np.stack([r['x'] for r in results]),
                fun=np.array([r['fun'] for r in results]),
                nit=np.array([r['nit'] for r in results]),
                nfev=np.array([r['nfev'] for r in results]),
                status=np.array([r['status'] for r in results]),
                hess_inv=[r['hess_inv'] for r in results], rounds=rounds)
