"""second_minimizer of vel_fit.process (vel_fit.py:653-658): scipy's BFGS for S
spectra in lock-step.  The per-spectrum state machines are C++20 coroutines
(csrc/bfgs_host.cpp, rvs_bfgs_*): scipy's `_minimize_bfgs` with its Wolfe line
searches (MINPACK-2 dcsrch / dcstep, the `line_search_wolfe2` fallback),
ScalarFunction's value caching and the 2-point gradient, each run suspended
where it needs function values; this driver gathers the requests of all runs
into one objective batch per round.

The Python statement of the same algorithm that the CPU suite pins to scipy
itself (identical nit / nfev / iterates) lives with the tests:
tests/refmachines/bfgs_scipy_restated.py; tests/test_tools_cpu.py compares the
C++ machines with it.
"""
import numpy as np


def minimize_lockstep_native(func, x0, hess_inv0=None, max_rows=None, gtol=1e-5,
                             c1=1e-4, c2=0.9, xrtol=0, maxiter=None):
    """minimize_lockstep with the per-spectrum state machines in C++
    (csrc/bfgs_host.cpp, rvs_bfgs_*): the same algorithm, scalar arithmetic in
    index order, so it follows the Python/scipy iterates to rounding rather than
    to the bit; the host cost per request drops from ~20 us to ~0.1 us."""
    import ctypes
    from . import _lib
    L = _lib.lib()
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    S, n = x0.shape
    H0 = None if hess_inv0 is None else np.ascontiguousarray(hess_inv0,
                                                             dtype=np.float64)

    def p(a):
        return None if a is None else a.ctypes.data_as(ctypes.c_void_p)

    h = L.rvs_bfgs_begin(S, n, p(x0), p(H0), float(gtol), float(c1), float(c2),
                         float(xrtol), int(maxiter or 0))
    if not h:
        raise ValueError('rvs_bfgs_begin: bad arguments (n <= 16)')
    h = ctypes.c_void_p(h)
    try:
        cap = S * (n + 1)
        idx = np.empty(cap, dtype=np.int64)
        X = np.empty((cap, n), dtype=np.float64)
        while True:
            rows = L.rvs_bfgs_pending(h, p(idx), p(X), cap)
            if rows < 0:
                raise RuntimeError('rvs_bfgs_pending failed (%d)' % rows)
            if rows == 0:
                break
            if max_rows is None or rows <= max_rows:
                F = np.asarray(func(idx[:rows], X[:rows]), dtype=np.float64)
            else:
                F = np.concatenate([
                    np.asarray(func(idx[a:min(rows, a + max_rows)],
                                    X[a:min(rows, a + max_rows)]),
                               dtype=np.float64)
                    for a in range(0, rows, max_rows)])
            F = np.ascontiguousarray(F)
            _lib.check(L.rvs_bfgs_feed(h, p(F), rows), 'rvs_bfgs_feed')
        x = np.empty((S, n))
        fun = np.empty(S)
        nit = np.empty(S, dtype=np.int32)
        nfev = np.empty(S, dtype=np.int32)
        status = np.empty(S, dtype=np.int32)
        Hk = np.empty((S, n, n))
        rounds = ctypes.c_int64(0)
        _lib.check(L.rvs_bfgs_result(h, p(x), p(fun), p(nit), p(nfev), p(status),
                                     p(Hk), ctypes.byref(rounds)),
                   'rvs_bfgs_result')
    finally:
        L.rvs_bfgs_end(h)
    return dict(x=x, fun=fun, nit=nit.astype(np.int64),
                nfev=nfev.astype(np.int64), status=status.astype(np.int64),
                hess_inv=list(Hk), rounds=int(rounds.value))
