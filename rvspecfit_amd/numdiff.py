"""Finite-difference Hessians as vel_fit.process asks numdifftools for them
(vel_fit.py:699-725), for a whole batch of spectra at once.

The reference calls
    ndf.Hessian(f, step=ndf.MinStepGenerator(base_step=b))(x)      first try
    ndf.Hessian(f, step=None)(x)                                   retry
(numdifftools 0.9.41, 'central' method).  What those calls compute:

* first try: MinStepGenerator's default number of steps for a central Hessian is
  max((n + order - 1) // 2, 1) + num_extrap = 1, so there is ONE step
  h = (b * s + 1) - 1 with s = max(log1p|x|, 1) and no extrapolation: the central
  rule  H_ii = (f(x+2h_i) - 2 f(x) + f(x-2h_i)) / (4 h_i^2),
        H_ij = (f(++) - f(+-) - f(-+) + f(--)) / (4 h_i h_j).
* retry: MaxStepGenerator(base_step=None, step_ratio=None): 15 steps
  h_k = EPS**(1/500) * s * 1.6**-k; the 15 central estimates are Richardson-
  extrapolated (three consecutive steps, error terms h^2 and h^4), passed once
  through Wynn's epsilon algorithm, and for every matrix element the estimate with
  the smallest error bound (after outlier trimming) is taken.

Every function evaluation of every spectrum of a stage goes through ONE batched
objective call per displacement pattern (`func(idx, points)`), i.e. through
rvs_chisq_point / rvs_objective_fused on the device; the extrapolation itself is
a few hundred flops per spectrum and runs on the host, all spectra at once.
"""
import numpy as np
import torch
from scipy.ndimage import convolve1d

EPS = float(np.finfo(float).eps)
TINY = float(np.finfo(float).tiny)
MAX_NUM_STEPS = 15
MAX_SCALE = 500.
HESS_STEP_RATIO = 1.6       # numdifftools' default ratio for derivatives of order > 1


def nominal_step(x):
    return torch.clamp(torch.log1p(x.abs()), min=1.0)


def first_try_step(base_step, x):
    """[S, n] the single step of MinStepGenerator(base_step=...) at x [S, n]"""
    return (base_step[None, :] * nominal_step(x) + 1.0) - 1.0


def retry_steps(x):
    """[15, S, n] the steps of the default generator, largest first"""
    base = EPS**(1. / MAX_SCALE) * nominal_step(x)
    k = torch.arange(MAX_NUM_STEPS, dtype=torch.float64, device=x.device)
    return base[None] * (HESS_STEP_RATIO**(-k))[:, None, None]


def _displacements(n, device):
    """the evaluation points of the central rule as signed multiples of the
    step vector: row 0 = x itself, then (+2e_i, -2e_i) per i, then
    (++, +-, -+, --) per pair i < j.  [M, n] float64, M = 1 + 2n + 2n(n-1)."""
    rows = [[0.0] * n]
    for i in range(n):
        for sg in (2.0, -2.0):
            r = [0.0] * n
            r[i] = sg
            rows.append(r)
    for i in range(n):
        for j in range(i + 1, n):
            for si, sj in ((1., 1.), (1., -1.), (-1., 1.), (-1., -1.)):
                r = [0.0] * n
                r[i], r[j] = si, sj
                rows.append(r)
    return torch.as_tensor(rows, dtype=torch.float64, device=device)


def hessian_central(func, x, h, max_rows=1 << 18):
    """Central second differences of func at x [S, n] with steps h [..., S, n]
    (leading axes = several step sizes per point): [..., S, n, n].
    func(idx, points) -> [len(idx)] evaluates rows idx.  All displacement
    patterns of all points (and step sizes) go through func in as few calls as
    max_rows allows -- one launch set instead of 1 + 2n + 2n(n-1) per step size;
    every function value is what the one-pattern-at-a-time loop computed (the
    objective kernels do not depend on the batch a row sits in)."""
    lead = h.shape[:-2]
    S, n = x.shape
    dev = x.device
    hh = h.reshape(-1, S, n)                       # [K, S, n]
    K = hh.shape[0]
    D = _displacements(n, dev)                     # [M, n]
    M = D.shape[0]
    # x + d * h with d in {0, +-1, +-2}: the products are exact, one rounding in
    # the sum, as x + 2 * ee[i] / x + ee[i] + ee[j] up to the order of two adds --
    # the pair patterns add h_i and h_j in two steps like numdifftools
    idx = torch.arange(S, device=dev)
    f = torch.empty((K, S, M), dtype=torch.float64, device=dev)
    per = max(1, max_rows // (S * M))
    for k0 in range(0, K, per):
        k1 = min(K, k0 + per)
        hk = hh[k0:k1]                             # [k, S, n]
        pts = x[None, :, None, :].expand(k1 - k0, S, M, n).clone()
        for c in range(n):       # (x + d_i h_i) + d_j h_j, i < j, like the loop
            pts[..., c] = pts[..., c] + D[None, None, :, c] * hk[:, :, None, c]
        rows = idx[None, :, None].expand(k1 - k0, S, M).reshape(-1)
        f[k0:k1] = func(rows, pts.reshape(-1, n)).reshape(k1 - k0, S, M)
    H = torch.empty((K, S, n, n), dtype=torch.float64, device=dev)
    fx = f[:, :, 0]
    m = 1
    for i in range(n):
        H[:, :, i, i] = (f[:, :, m] - 2 * fx + f[:, :, m + 1]) / \
            (4. * hh[:, :, i] * hh[:, :, i])
        m += 2
    for i in range(n):
        for j in range(i + 1, n):
            v = (f[:, :, m] - f[:, :, m + 1] - f[:, :, m + 2] + f[:, :, m + 3]) / \
                (4. * hh[:, :, i] * hh[:, :, j])
            H[:, :, i, j] = v
            H[:, :, j, i] = v
            m += 4
    return H.reshape(lead + (S, n, n))


# ---- extrapolation of a sequence of estimates (host, all columns at once) ----
def _richardson_rule(step_ratio, nterms=2, step=2, order=2):
    i, j = np.ogrid[0:nterms + 1, 0:nterms]
    r = np.ones((nterms + 1, nterms + 1))
    r[:, 1:] = (1.0 / step_ratio)**(i * (step * j + order))
    return np.linalg.pinv(r)[0]


def _richardson(seq, step_ratio):
    """seq [K, M], K >= 3 -> extrapolated [K-2, M] and error bounds"""
    K = seq.shape[0]
    rule = _richardson_rule(step_ratio)
    m = K - 2
    new = convolve1d(seq, rule[::-1], axis=0, origin=1)[:m + 1]
    fact = max(12.7062047361747 * np.sqrt(np.sum(rule**2)), EPS * 10.)
    err = np.abs(np.diff(new, axis=0)) * fact
    tol = np.maximum(np.abs(new[1:]), np.abs(new[:-1])) * EPS * fact
    abserr = err + np.where(err <= tol, tol * 10,
                            np.abs(new[:-1] - seq[1:m + 1]) * fact)
    return new[:m], abserr[:m]


def _wynn(seq):
    """one pass of Wynn's epsilon algorithm over consecutive triples"""
    e0, e1, e2 = seq[:-2], seq[1:-1], seq[2:]
    with np.errstate(all='ignore'):
        d2, d1 = e2 - e1, e1 - e0
        err2, err1 = np.abs(d2), np.abs(d1)
        tol2 = np.maximum(np.abs(e2), np.abs(e1)) * EPS
        tol1 = np.maximum(np.abs(e1), np.abs(e0)) * EPS
        d1 = np.where(err1 < TINY, TINY, d1)
        d2 = np.where(err2 < TINY, TINY, d2)
        ss = 1.0 / d2 - 1.0 / d1 + TINY
        conv = ((err1 <= tol1) & (err2 <= tol2)) | (np.abs(ss * e1) <= 1.0e-3)
        res = np.where(conv, e2, e1 + 1.0 / ss)
        abserr = err1 + err2 + np.where(conv, tol2 * 10, np.abs(res - e2))
    return res, abserr


def _nanpercentile_cols(a, q):
    """np.nanpercentile(a, q, axis=0) (method 'linear') for every column at once.
    numpy walks the columns one by one in Python (apply_along_axis: 20 us per
    column, 0.6 s of a 10 000-spectra Hessian stage); here one sort puts the NaNs
    of a column last, the virtual index (n_valid - 1) q / 100 is formed per
    column and the two neighbours are blended with numpy's own `_lerp`
    (a + (b - a) t, and b - (b - a)(1 - t) for t >= 0.5): the same values."""
    a = np.asarray(a, dtype=np.float64)
    srt = np.sort(a, axis=0)                      # NaN sorts to the end
    n = np.sum(~np.isnan(a), axis=0)
    vi = (n - 1) * (q / 100.0)
    lo = np.floor(vi)
    t = vi - lo
    lo = np.clip(lo, 0, max(a.shape[0] - 1, 0)).astype(np.int64)
    hi = np.clip(lo + 1, 0, np.maximum(n - 1, 0)).astype(np.int64)
    cols = np.arange(a.shape[1])
    x, y = srt[lo, cols], srt[hi, cols]
    d = y - x
    out = x + d * t
    out = np.where(t >= 0.5, y - d * (1 - t), out)
    # numpy: lerp(a, b, 0) = a exactly -- for a finite difference only (with an
    # infinite neighbour numpy's inf * 0 is NaN, and so is this)
    out = np.where((t == 0) & np.isfinite(d), x, out)
    return np.where(n == 0, np.nan, out)


def _outlier_errors(der, trim_fact=10):
    with np.errstate(all='ignore'):
        med = np.nanmedian(der, axis=0)
        p75 = _nanpercentile_cols(der, 75)
        p25 = _nanpercentile_cols(der, 25)
        iqr = np.abs(p75 - p25)
        am = np.abs(med)
        out = (((np.abs(der) < am / trim_fact) | (np.abs(der) > am * trim_fact))
               & (am > 1e-8)) | (der < p25 - 1.5 * iqr) | (p75 + 1.5 * iqr < der)
        return out * np.abs(der - med)


def extrapolate(seq, step_ratio=HESS_STEP_RATIO):
    """seq [K, ...]: K estimates at steps shrinking by step_ratio -> the best
    extrapolated estimate per trailing element"""
    shape = seq.shape[1:]
    s2 = np.asarray(seq, dtype=np.float64).reshape(seq.shape[0], -1)
    if s2.shape[0] < 3:
        return s2[-1].reshape(shape)
    der, err = _richardson(s2, step_ratio)
    if der.shape[0] > 2:
        der, err = _wynn(der)
    err = err + _outlier_errors(der)
    # per column: the row of the smallest error bound, the middle one of equal
    # minima (np.flatnonzero(col == nanmin)[size // 2]); all-NaN columns -> NaN
    # (numdifftools falls back to an arbitrary row there)
    allnan = np.isnan(err).all(axis=0)
    with np.errstate(all='ignore'):
        mn = np.nanmin(np.where(allnan[None, :], 0.0, err), axis=0)
    hit = (err == mn[None, :])
    rank = np.cumsum(hit, axis=0)                 # 1-based rank among the hits
    want = hit.sum(axis=0) // 2 + 1
    pick = np.argmax(hit & (rank == want[None, :]), axis=0)
    out = der[pick, np.arange(der.shape[1])]
    out = np.where(allnan, np.nan, out)
    return out.reshape(shape)


def hessian_retry(func, x):
    """ndf.Hessian(f)(x) with the default step generator, rows of x [S, n]"""
    return extrapolate(hessian_central(func, x, retry_steps(x)).cpu().numpy())
