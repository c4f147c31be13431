"""Host-side, once-per-arm tables for the CCF kernels (numpy only).

Everything here depends only on the arm's wavelength grid and the CCF
configuration -- never on a spectrum -- so it is computed once on the host in
float64/int64 exactly as the reference computes it, and uploaded:

  * the lag -> velocity tables and the sub-index of lags inside +-max_vel
    (fitter_ccf.py:132-161) -- integer work, bit-exact by construction;
  * the 2-point rebin tables onto the log-lambda FFT grid (make_ccf.py:394-404);
  * the pixel ranges of the binned-median start of the continuum fit
    (make_ccf.py:128-143, scipy.stats.binned_statistic semantics);
  * the design matrix of the k=2 INTERPOLATING spline through the continuum
    nodes (make_ccf.py:155-164: UnivariateSpline(nodes, p, s=0, k=2)(lam)).
    An interpolating spline is linear in its node values p, so
    spline(lam) = Lmat @ p.  Lmat is built from the same FITPACK knot rule
    (fpcurf.f, s=0, even k: interior knots at the mid points of the data
    abscissae) with B-spline collocation; extrapolation beyond the last node
    uses the end polynomial piece (ext=0).
"""
import numpy as np


def _bspl_basis(t, k, x, l):
    """The k+1 B-splines of degree k that are non zero on [t[l], t[l+1]),
    evaluated at x (polynomially extended outside): FITPACK fpbspl."""
    h = np.zeros(k + 1)
    h[0] = 1.0
    for j in range(1, k + 1):
        hh = h[:j].copy()
        h[0] = 0.0
        for i in range(j):
            li = l + i + 1
            lj = li - j
            f = hh[i] / (t[li] - t[lj])
            h[i] += f * (t[li] - x)
            h[i + 1] = f * (x - t[lj])
    return h


def interp_spline_tables(nodes, lam, k=2):
    """B-spline form of UnivariateSpline(nodes, p, s=0, k=2)(lam):
        S(lam_r) = sum_q Eb[r, q] * (Cinv @ p)[El[r] + q],  q = 0..2
    Returns Eb [npix, 3], El int32 [npix] (non decreasing), Cinv [m, m] and
    istart int32 [m-1]: first pixel of every knot interval (m-2 intervals)."""
    nodes = np.asarray(nodes, dtype=np.float64)
    lam = np.asarray(lam, dtype=np.float64)
    m = len(nodes)
    assert m > k and k == 2
    # FITPACK knots for s=0, k=2 (fpcurf.f): interior knots at data mid points
    interior = 0.5 * (nodes[1:m - 2] + nodes[2:m - 1])
    t = np.concatenate([[nodes[0]] * 3, interior, [nodes[-1]] * 3])
    n = len(t)
    assert n == m + k + 1

    def basis(xs):
        # FITPACK fpbspl for k = 2, all abscissae at once (the operations of
        # _bspl_basis in the same order: the same values)
        xs = np.asarray(xs, dtype=np.float64)
        l = np.searchsorted(t, xs, 'right') - 1
        l = np.minimum(np.maximum(l, k), n - k - 2)   # ext=0: end pieces extrapolate
        tl, tl1, tl2, tlm1 = t[l], t[l + 1], t[l + 2], t[l - 1]
        # j = 1
        f = 1.0 / (tl1 - tl)
        a0 = f * (tl1 - xs)
        a1 = f * (xs - tl)
        # j = 2
        f = a0 / (tl1 - tlm1)
        h0 = f * (tl1 - xs)
        h1 = f * (xs - tlm1)
        f = a1 / (tl2 - tl)
        h1 = h1 + f * (tl2 - xs)
        h2 = f * (xs - tl)
        Eb = np.stack([h0, h1, h2], axis=1)
        return Eb, (l - k).astype(np.int32)

    Cb, Cl = basis(nodes)
    C = np.zeros((m, m))
    for r in range(m):
        C[r, Cl[r]:Cl[r] + 3] = Cb[r]
    Eb, El = basis(lam)
    assert np.all(np.diff(El) >= 0)
    nint = m - 2
    istart = np.searchsorted(El, np.arange(nint + 1)).astype(np.int32)
    istart[nint] = len(lam)
    return Eb, El, np.linalg.inv(C), istart


def collocation(nodes, k=2):
    """(C^-1, C) of the interpolating spline through `nodes` alone: the part of
    interp_spline_tables that does not depend on the pixels (the pixel part is
    rvs_ccf_tables_build's)"""
    _, _, Cinv, _ = interp_spline_tables(nodes, np.asarray(nodes)[:1], k)
    return Cinv, np.linalg.inv(Cinv)


def collocation_batch(nodes, m):
    """collocation() for G node sets at once.  nodes [G, nn] (row g: m[g] nodes,
    then padding), m int [G].  Returns [G, 2 nn nn]: row g holds C^-1 then C of its
    m[g] x m[g] collocation matrix, packed.  The interval of node r is known in
    closed form -- the knots are the mid points of the nodes, so
    searchsorted(t, node_r, 'right') - 1, clamped, is clamp(r + 1, 2, m - 1) -- the
    basis values are interp_spline_tables' arithmetic, and the stacked inverses run
    LAPACK matrix by matrix: the same bits as grid by grid."""
    nodes = np.asarray(nodes, dtype=np.float64)
    m = np.asarray(m, dtype=np.int64)
    G, nn = nodes.shape
    k = 2
    r = np.arange(nn)[None, :]
    valid = r < m[:, None]
    last = np.take_along_axis(nodes, (m - 1)[:, None], axis=1)
    # knots t [G, nn + 3]: node 0 three times, interior mid points, last node x 3
    j = np.arange(nn + 3)[None, :]
    ia = np.clip(j - 2, 0, nn - 1)
    ib = np.clip(j - 1, 0, nn - 1)
    t = 0.5 * (np.take_along_axis(nodes, np.broadcast_to(ia, (G, nn + 3)), axis=1) +
               np.take_along_axis(nodes, np.broadcast_to(ib, (G, nn + 3)), axis=1))
    t = np.where(j < 3, nodes[:, :1], t)
    t = np.where(j >= m[:, None], last, t)
    l = np.clip(r + 1, k, (m - 1)[:, None])
    l = np.where(valid, l, k)

    def tk(off):
        return np.take_along_axis(t, np.clip(l + off, 0, nn + 2), axis=1)
    xs = np.where(valid, nodes, tk(0))
    tl, tl1, tl2, tlm1 = tk(0), tk(1), tk(2), tk(-1)
    with np.errstate(all='ignore'):
        f = 1.0 / (tl1 - tl)
        a0 = f * (tl1 - xs)
        a1 = f * (xs - tl)
        f = a0 / (tl1 - tlm1)
        h0 = f * (tl1 - xs)
        h1 = f * (xs - tlm1)
        f = a1 / (tl2 - tl)
        h1 = h1 + f * (tl2 - xs)
        h2 = f * (xs - tl)
    C = np.zeros((G, nn, nn))
    gi = np.arange(G)[:, None]
    ri = np.broadcast_to(r, (G, nn))
    for q, h in enumerate((h0, h1, h2)):
        col = np.clip(l - k + q, 0, nn - 1)
        C[gi, ri, col] = np.where(valid, h, C[gi, ri, col])
    # LAPACK matrix by matrix, the grids of one node count stacked (a padded matrix
    # would be factored with other recursion splits: not the same bits)
    out = np.zeros((G, 2 * nn * nn))
    for mm in np.unique(m):
        sel = np.nonzero(m == mm)[0]
        Ci = np.linalg.inv(np.ascontiguousarray(C[sel, :mm, :mm]))
        out[sel, :mm * mm] = Ci.reshape(len(sel), -1)
        out[sel, mm * mm:2 * mm * mm] = np.linalg.inv(Ci).reshape(len(sel), -1)
    return out


def interp_spline_design(nodes, lam, k=2):
    """Dense Lmat [len(lam), len(nodes)] with
    UnivariateSpline(nodes, p, s=0, k=2)(lam) == Lmat @ p (tests)."""
    Eb, El, Cinv, _ = interp_spline_tables(nodes, lam, k)
    E = np.zeros((len(lam), len(nodes)))
    for r in range(len(lam)):
        E[r, El[r]:El[r] + 3] = Eb[r]
    return E @ Cinv


def continuum_nodes(lam0, splinestep):
    """make_ccf.py:123-131"""
    lammin = lam0.min()
    dl = np.log(1 + splinestep / 3e5)
    N = int(np.ceil(np.log(lam0.max() / lammin) / dl))
    nodes = lammin * np.exp(np.arange(N) * dl)
    edges = lammin * np.exp((-0.5 + np.arange(N + 1)) * dl)
    return nodes, edges


def bin_ranges(lam, edges):
    """Pixel ranges [start_j, start_{j+1}) of every bin of
    scipy.stats.binned_statistic(lam, ., bins=edges) for an increasing lam:
    bin j holds edges[j] <= lam < edges[j+1], the last bin is closed on the
    right, pixels outside the edges belong to no bin."""
    lam = np.asarray(lam)
    nb = len(edges) - 1
    start = np.searchsorted(lam, edges[:-1], 'left').astype(np.int32)
    last_end = np.searchsorted(lam, edges[-1], 'right')
    out = np.zeros(nb + 1, dtype=np.int32)
    out[:nb] = start
    out[nb] = last_end
    # first pixel of a bin can not precede the previous bin's start
    return out


def rebin_tables(lam, logl0, logl1, npoints):
    """make_ccf.py:355-357, 394-399: xind (-1 where the FFT grid point has no
    bracketing pixel pair) and the right weight."""
    ccf_lam = np.exp(np.linspace(logl0, logl1, npoints))
    xind = np.searchsorted(lam, ccf_lam) - 1
    sub = (xind >= 0) & (xind <= len(lam) - 2)
    rw = np.zeros(npoints)
    li = xind[sub]
    rw[sub] = (ccf_lam[sub] - lam[li]) / (lam[li + 1] - lam[li])
    xi = np.where(sub, xind, -1).astype(np.int32)
    return xi, rw


def lag_tables(logl0, logl1, npoints, maxvel):
    """fitter_ccf.py:132-154"""
    step = (np.exp((logl1 - logl0) / npoints) - 1) * 3e5
    L = npoints
    off = L // 2
    vels = -((np.arange(L) + off) % L - off) * step
    sel = np.abs(vels) < (maxvel + step)
    assert sel.sum() % 2 == 1
    ind = np.roll(np.nonzero(sel)[0], sel.sum() // 2)[::-1]
    sub = vels[ind]
    if not np.all(np.diff(sub) > 0):
        raise RuntimeError('Velocity grid for CCF interpolation is invalid')
    return step, ind.astype(np.int32), np.ascontiguousarray(sub)


def interp_tables(sub_vels, vel_grid):
    """interp1d(kind='linear', assume_sorted=True) bracketing indices."""
    hi = np.clip(np.searchsorted(sub_vels, vel_grid), 1, len(sub_vels) - 1)
    return (hi - 1).astype(np.int32)


def ccf_vel_grid(config):
    """fitter_ccf.py:81-87"""
    maxvel = config.get('max_vel') or 1000
    nvel = 2 * int(maxvel * 1. / (config.get('vel_step0') or 2)) + 1
    return maxvel, np.linspace(-maxvel, maxvel, nvel)
