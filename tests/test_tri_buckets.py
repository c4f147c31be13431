"""find_simplex of Delaunay libraries through a bucket grid
(rvs_template_tri_buckets, library.tri_buckets) against the exhaustive search: the
same simplex id -- the lowest one that passes scipy's inside test
(spec_inter.py:11-59, Delaunay.find_simplex with eps = 100 DBL_EPSILON) -- for
every point."""
import os

import numpy as np
import pytest

from conftest import GOLD

EPS = 100 * 2.220446049250313e-16


def _inside(T, p, nd):
    c = np.einsum('sij,sj->si', T[:, :nd, :], p[None, :] - T[:, nd, :])
    cl = 1 - c.sum(1)
    return (c >= -EPS).all(1) & (c <= 1 + EPS).all(1) & (cl >= -EPS) & (cl <= 1 + EPS)


def _grid_delaunay(n, jitter, seed):
    import scipy.spatial
    rng = np.random.RandomState(seed)
    g = [np.linspace(0, 1, n)] * 4
    pts = np.array(np.meshgrid(*g, indexing='ij')).reshape(4, -1).T
    pts = pts + jitter * rng.uniform(-1, 1, pts.shape) / (n - 1)
    return pts, scipy.spatial.Delaunay(pts)


@pytest.mark.parametrize('jitter,wide', [(0.0, 10**9), (0.2, 10**9), (0.2, 20)])
def test_bucket_lists_hold_every_candidate(jitter, wide):
    """host side: for random points (inside, on the hull, outside) the lowest
    matching simplex of the point's cell list is the lowest matching simplex of the
    whole triangulation; lists ascending; a regular grid (Qhull's triangulated
    cospherical facets, what a PHOENIX grid gives) and a jittered one; wide = 20: the
    simplices whose boxes overlap more than 20 cells in the list of their own that
    every query tests as well"""
    from rvspecfit_amd.library import tri_buckets
    pts, D = _grid_delaunay(5, jitter, 3)
    nd = 4
    bk = tri_buckets(D.transform, nd, wide_cells=wide)
    st, ls = bk['cell_start'], bk['cell_list']
    ncell = int(np.prod(bk['n']))
    assert len(st) == ncell + 2 and st[-1] == len(ls) and np.all(np.diff(st) >= 0)
    for c in range(len(st) - 1):
        assert np.all(np.diff(ls[st[c]:st[c + 1]]) > 0)
    wide_list = ls[st[ncell]:st[ncell + 1]]
    assert (len(wide_list) > 0) == (wide < 10**6)
    rng = np.random.RandomState(5)
    q = np.concatenate([rng.uniform(-0.05, 1.05, (600, 4)), pts[::7],
                        0.5 * (pts[10:200:3] + pts[11:201:3])])
    for p in q:
        m = np.nonzero(_inside(D.transform, p, nd))[0]
        want = m[0] if len(m) else -1
        c = np.clip(np.floor((p - bk['lo']) * bk['inv_w']), 0, bk['n'] - 1).astype(int)
        cell = 0
        for d in range(nd):
            cell = cell * bk['n'][d] + c[d]
        got = []
        for lst in (ls[st[cell]:st[cell + 1]], wide_list):
            mm = lst[_inside(D.transform[lst], p, nd)] if len(lst) else []
            if len(mm):
                got.append(mm[0])
        assert (min(got) if got else -1) == want
    assert np.diff(st)[:ncell].mean() < 0.25 * len(D.simplices)


def _tri_library(pts, D, ntp=40, seed=2):
    from rvspecfit_amd.library import TemplateLibrary
    g = dict(np.load(os.path.join(GOLD, 'lib_tri_gold_b.npz')))
    rng = np.random.RandomState(seed)
    lam = np.exp(np.linspace(np.log(4000.), np.log(4100.), ntp))
    d = dict(lam=lam, dats=0.1 * rng.normal(size=(len(pts), ntp)), vec=pts.T,
             simplices=D.simplices.astype(np.int32), transform=D.transform,
             extraflags=np.zeros(len(pts)), log_step=np.array(True),
             log_ids=np.array([], dtype=np.int64), parnames=g['parnames'],
             interpolation_type=g['interpolation_type'])
    return TemplateLibrary('tri_test', d)


@pytest.mark.gpu
@pytest.mark.parametrize('n,jitter,wide', [(7, 0.0, 4096), (7, 0.25, 4096),
                                           (7, 0.25, 24)])
def test_bucket_search_equals_exhaustive_on_the_device(n, jitter, wide, monkeypatch):
    """10^5 random points (a tenth outside the hull), 31 000 simplices: the ids of
    rvs_template_tri_buckets are those of rvs_template_tri, the templates too; wide =
    24: thousands of simplices in the list every query tests beside its cell's"""
    import torch
    from rvspecfit_amd import library
    monkeypatch.setattr(library, 'TRI_WIDE_CELLS', wide)
    pts, D = _grid_delaunay(n, jitter, 11)
    assert len(D.simplices) > 20000
    lib = _tri_library(pts, D)
    assert lib._tri_bk is not None
    assert (lib.tri_nwide > 100) == (wide < 100), lib.tri_nwide
    rng = np.random.RandomState(9)
    q = rng.uniform(-0.03, 1.03, (100000, 4))
    q[:2000] = pts[rng.randint(0, len(pts), 2000)]          # vertices
    q[2000:4000] = 0.5 * (pts[rng.randint(0, len(pts), 2000)] +
                          pts[rng.randint(0, len(pts), 2000)])
    qt = torch.as_tensor(q).to('cuda')
    t1, o1, s1, w1 = lib.eval_batch(qt, details=True)
    library.TRI_BUCKETS = False
    try:
        t0, o0, s0, w0 = lib.eval_batch(qt, details=True)
    finally:
        library.TRI_BUCKETS = True
    assert torch.equal(s0, s1)
    found = (s0 != 0x7fffffff)
    assert found.float().mean().item() > 0.7
    assert torch.equal(w0, w1)
    assert torch.equal(t0[found], t1[found])
    assert torch.isnan(t1[~found]).all()
