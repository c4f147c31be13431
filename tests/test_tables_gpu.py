"""The per-grid tables built on the device (csrc/tables.hip: rvs_basis_build,
rvs_ccf_tables_build) against the numpy statements they replace (engine.get_poly_basis
+ np.linalg.qr, rvspecfit_amd/ccf_tables.py), on a grid set of SDSS-shaped pieces and
on the DESI arms."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _arm(grids, rng):
    from rvspecfit_amd.engine import ArmData
    G = len(grids)
    npx = max(len(g) for g in grids)
    sp = rng.normal(1, 0.1, (G, npx))
    es = np.full((G, npx), 0.05)
    if G == 1:
        return ArmData('t', grids[0], sp, es)
    return ArmData('t', grids, sp, es, grid_id=np.arange(G, dtype=np.int32))


def _both(fn):
    from rvspecfit_amd import engine
    keep = engine.DEVICE_TABLES
    out = []
    try:
        for flag in (True, False):
            engine.DEVICE_TABLES = flag
            out.append(fn())
    finally:
        engine.DEVICE_TABLES = keep
    return out


GRIDS = {
    'sdss_pieces': lambda: [(10**(3.5798 + 1e-4 * np.arange(3842)))[a:a + n]
                            for a, n in ((0, 3842), (37, 3805), (255, 3200),
                                         (800, 3042), (5, 3500))],
    'desi_b': lambda: [np.arange(3600., 5800.1, 0.8)],
    'short': lambda: [np.linspace(4000., 5000., 2001), np.linspace(4100., 4900., 700)],
    # beyond the 8192 pixels one register set holds (64 pixels per thread)
    'long': lambda: [np.linspace(3600., 9800., 9000)],
    'long_set': lambda: [np.linspace(3600., 9800., 9300), np.linspace(3700., 9000., 8200)],
}


@pytest.mark.parametrize('name', list(GRIDS))
@pytest.mark.parametrize('npoly,rbf', [(10, True), (15, True), (3, True), (2, True),
                                       (7, False), (15, False), (1, False),
                                       # what rvs_chisq_full / _continuum take
                                       # beyond the grid kernel's 16
                                       (17, True), (24, False), (32, True)])
def test_basis(name, npoly, rbf):
    grids = GRIDS[name]()
    rng = np.random.RandomState(2)

    def run():
        a = _arm(grids, rng)
        raw = a.basis(npoly, rbf).cpu().numpy().reshape(len(grids), a.npix + 1, npoly)
        qt, off = a.basis_ortho(npoly, rbf)
        off = off.cpu().numpy() if torch.is_tensor(off) else np.array([off])
        return raw, qt.cpu().numpy().reshape(raw.shape), off
    (rd, qd, od), (rh, qh, oh) = _both(run)
    for i, g in enumerate(grids):
        n = len(g)
        assert not rd[i, n:].any() and not qd[i, n:].any()
        if rbf and npoly > 3:   # the Gaussians: device exp vs numpy exp, an ulp
            assert np.array_equal(rd[i, :n, :3], rh[i, :n, :3])
            assert np.abs(rd[i, :n] - rh[i, :n]).max() <= 4e-16
        else:
            assert np.array_equal(rd[i, :n], rh[i, :n])
        # orthonormal, the same space, the same volume
        Q, H = qd[i, :n], qh[i, :n]
        assert np.abs(Q.T @ Q - np.eye(npoly)).max() < 1e-13
        assert np.abs(H @ (H.T @ Q) - Q).max() < (1e-10 if npoly <= 16 else 1e-8)
        assert abs(od[i] - oh[i]) <= 1e-10 * max(1.0, abs(oh[i]))


@pytest.mark.parametrize('name', ['sdss_pieces', 'desi_b'])
@pytest.mark.parametrize('continuum', [True, False])
def test_ccf_tables(name, continuum):
    from rvspecfit_amd import ccf_tables

    class Lib:   # what ArmData.ccf_tables reads of a library
        name = 'fake'

        def __init__(self, lo, hi):
            self.cc = dict(npoints=4096, logl0=np.log(lo), logl1=np.log(hi),
                           continuum=continuum, splinestep=max(1000., 3e5 * (
                               np.exp(np.log(hi / lo) / 20) - 1)))

        def ccf_set(self, config):
            return self.cc
    grids = GRIDS[name]()
    lib = Lib(min(g[0] for g in grids) * 1.01, max(g[-1] for g in grids) * 0.99)
    cfg = dict(max_vel=1000, vel_step0=5)
    rng = np.random.RandomState(4)
    td, th = _both(lambda: _arm(grids, rng).ccf_tables(lib, cfg))
    keys = ['xind', 'rw'] + (['Eb', 'El', 'istart', 'bin_start', 'Cinv', 'nnode_g',
                              'npix_g'] if continuum else [])
    for k in keys:
        a, b = td[k], th[k]
        if a is None or b is None:
            assert a is None and b is None, k
            continue
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert a.shape == b.shape, k
        if k == 'Cinv':     # LAPACK both times
            assert np.array_equal(a, b)
        else:               # searches and IEEE arithmetic: the host's bits
            assert np.array_equal(a, b), (k, np.abs(a.astype(float) - b).max())
    assert td['nnode'] == th['nnode']


def test_raw_basis_alone_skips_the_orthonormalisation():
    """basis() asks rvs_basis_build for the raw table only (ortho = NULL); a later
    basis_ortho() builds both and the raw table is the same bits"""
    rng = np.random.RandomState(4)
    a = _arm(GRIDS['desi_b'](), rng)
    raw = a.basis(10, True).clone()
    assert ('devraw', 10, True) in a._basis and ('dev', 10, True) not in a._basis
    qt, off = a.basis_ortho(10, True)
    assert ('dev', 10, True) in a._basis and ('devraw', 10, True) not in a._basis
    assert torch.equal(a.basis(10, True), raw)
    assert np.isfinite(off) and qt.shape == raw.shape
