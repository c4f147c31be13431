"""second_minimizer of vel_fit.process (vel_fit.py:653-658): scipy's BFGS for S
spectra in lock-step.  The per-spectrum state machine (csrc/bfgs_machine.h) is
scipy's `_minimize_bfgs` with its Wolfe line searches (MINPACK-2 dcsrch / dcstep,
the `line_search_wolfe2` fallback), ScalarFunction's value caching and the
2-point gradient, each run suspended where it needs function values.  Two
drivers around the one machine:
  minimize_lockstep_device  the rounds on the GPU (rvs_bfgs_run: one thread per
                            run, the requests of all runs gathered into one
                            objective batch per round by kernels) -- what
                            vel_fit.process takes wherever the library launches
                            the objective itself (regular-grid and MLP libraries)
  minimize_lockstep_native  the machines on the host (rvs_bfgs_begin / _pending /
                            _feed) around any Python objective

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


def minimize_lockstep_device(pobj, x0, hess_inv0=None, gtol=1e-5, c1=1e-4, c2=0.9,
                             xrtol=0, maxiter=None, sync_every=4):
    """The runs on the device (csrc/bfgs_dev.hip, rvs_bfgs_run) around an
    optimizer.ProcessObjective whose rounds the library drives (pobj.fused or
    pobj.nn_native): x0 [S, n] device tensor, hess_inv0 [n, n] array.  Returns
    device tensors x [S, n], fun, nit, nfev, status [S] and the statistics of the
    run (rounds, objective calls, rows launched)."""
    import ctypes
    import torch
    from . import _lib
    L = _lib.lib()
    dev = x0.device
    S, n = x0.shape
    if n != pobj.n or S != pobj.S:
        raise ValueError('minimize_lockstep_device: x0 does not fit the objective')
    f64 = dict(dtype=torch.float64, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    rows = S * (n + 1)
    x0 = x0.to(torch.float64).contiguous()
    H0 = None if hess_inv0 is None else torch.as_tensor(
        np.ascontiguousarray(hess_inv0, dtype=np.float64)).to(dev).contiguous()
    keep = dict(
        runs=torch.empty(S * int(L.rvs_bfgs_run_bytes()) // 8 + 1, **f64),
        x0=x0, hess_inv0=H0, x=torch.empty((S, n), **f64),
        fun=torch.empty(S, **f64), hess_inv=None,
        nit=torch.empty(S, **i32), nfev=torch.empty(S, **i32),
        status=torch.empty(S, **i32), nreq=torch.zeros(S, **i32),
        off=torch.zeros(S, **i32), list=torch.zeros(rows, **i32),
        counts=torch.zeros(32, **i32), X=torch.zeros((rows, n), **f64),
        F=torch.zeros(rows, **f64))
    b = _lib.BfgsState()
    for k, t in keep.items():
        setattr(b, k, None if t is None else t.data_ptr())
    b.gtol, b.c1, b.c2, b.xrtol = float(gtol), float(c1), float(c2), float(xrtol)
    b.S, b.n, b.cap, b.maxiter = S, n, int(pobj.cap), int(maxiter or 0)
    o = pobj.native_desc()
    st3 = (ctypes.c_int64 * 3)()
    _lib.check(L.rvs_bfgs_run(ctypes.addressof(b), ctypes.addressof(o),
                              int(sync_every), st3, _lib.stream()),
               'rvs_bfgs_run')
    pobj.calls += int(st3[1])
    nfev = keep['nfev'].long()
    pobj.jobs += int(nfev.sum().item())
    pobj.slots = getattr(pobj, 'slots', 0) + int(st3[2])
    return dict(x=keep['x'], fun=keep['fun'], nit=keep['nit'].long(), nfev=nfev,
                status=keep['status'].long(), rounds=int(st3[0]),
                calls=int(st3[1]), rows_launched=int(st3[2]))
