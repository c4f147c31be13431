"""Extended-precision evaluation of the continuum-marginalised -2 log L of
get_chisq0 (spec_fit.py:203-354), used to tell WHOSE rounding a difference
between the float64 evaluations belongs to (tests/test_chisq_accuracy.py).

-2 log L = log det(ST ST^T) + 2 sum log e + |D - ST^T a|^2,  ST = polys * t / e,
D = s / e, a = (ST ST^T)^-1 ST D.  Here: Householder QR of ST^T in np.longdouble
(x87 80-bit, eps 1.1e-19): log det = 2 sum log |R_ii|, residual = the part of D
orthogonal to the columns, formed explicitly."""
import numpy as np

LD = np.longdouble


def chisq0_longdouble(spec, templ, polys, espec):
    e = np.asarray(espec, dtype=LD)
    D = np.asarray(spec, dtype=LD) / e
    A = (np.asarray(polys, dtype=LD) * (np.asarray(templ, dtype=LD) / e)[None, :]).T
    n, p = A.shape
    A = A.copy()
    y = D.copy()
    logdet = LD(0)
    for k in range(p):
        x = A[k:, k]
        nx = np.sqrt(np.sum(x * x))
        alpha = -nx if x[0] > 0 else nx
        v = x.copy()
        v[0] -= alpha
        vv = np.sum(v * v)
        if vv > 0:
            A[k:, k:] -= np.outer(v, (2 / vv) * (v @ A[k:, k:]))
            y[k:] -= v * ((2 / vv) * (v @ y[k:]))
        logdet += 2 * np.log(np.abs(A[k, k]))
    resid = np.sum(y[p:] * y[p:])
    return logdet + 2 * np.sum(np.log(e)) + resid


def chisq0_orthonormal_f64(spec, templ, polys, espec):
    """float64 emulation of the DEVICE's arithmetic (csrc/chisq.hip): the basis
    is orthonormalised once (QR of polys^T; 2 log|det R| added back), the normal
    matrix of the orthonormal basis is Cholesky factorised and
    -2 log L = 2 sum log L_ii + 2 sum log e + (D.D - y.y), y = L^-1 v."""
    Q, R = np.linalg.qr(np.asarray(polys, dtype=np.float64).T)
    off = -2 * np.sum(np.log(np.abs(np.diag(R))))   # det(polys) -> det(Q)
    e = np.asarray(espec, dtype=np.float64)
    t = np.asarray(templ, dtype=np.float64)
    s = np.asarray(spec, dtype=np.float64)
    w = t * t / (e * e)
    u = t * s / (e * e)
    M = (Q * w[:, None]).T @ Q
    v = Q.T @ u
    L = np.linalg.cholesky(M)
    y = np.linalg.solve(L, v)   # forward substitution (small, dense)
    dd = np.sum((s / e)**2)
    # the device reports chi^2 in the ORIGINAL basis: log det(ST ST^T) =
    # log det(M_orth) + 2 log |det R|
    return 2 * np.sum(np.log(np.diag(L))) - off + 2 * np.sum(np.log(e)) + \
        (dd - y @ y)
