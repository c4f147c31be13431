"""Template grids whose dimensions have different lengths (9 x 6 x 5 x 3, unequal
node spacing, holes, log-mapped teff): tests/golden/ragged_cases.npz holds what the
reference's GridInterp / GridOutsideCheck / get_chisq / vel_fit.process return on
tests/golden/lib_rag_b.npz (objects built in memory by make_golden_ragged.py,
spec_inter.py:62-194, read_grid.py:114-145).

CPU: the oracle against those vectors.  GPU: rvs_template_polylinear (cell index,
branch, vertex ids bit-exact), the cell search of rvs_objective_fused, the
`generic` plug-in classes and vel_fit.process."""
import os

import numpy as np
import pytest

from conftest import GOLD
from oracle import rvs_oracle as orc
from test_numpy_expf import host_numpy_expf_is_published_algorithm

SETUP = 'rag_b'
CONFIG = dict(min_vel=-1000, max_vel=1000, min_vel_step=0.2, vel_step0=5,
              min_vsini=0.1, max_vsini=500, second_minimizer=False)


@pytest.fixture(scope='module')
def rag():
    return dict(np.load(os.path.join(GOLD, 'ragged_cases.npz')))


@pytest.fixture(scope='module')
def libdict():
    return dict(np.load(os.path.join(GOLD, 'lib_%s.npz' % SETUP)))


@pytest.fixture(scope='module')
def olib(libdict):
    return orc.Library(libdict)


def _sds(rag, cls):
    return [cls(SETUP, rag['spec/lam'], rag['spec/spec'], rag['spec/espec'],
                badmask=rag['spec/badmask'])]


def libm_sensitive(libdict, rag, i):
    """mapped teff within a few ulp of a node: log10 is third-party arithmetic
    at the boundary (the capture interpreter's libm rounds log10(6000) up, glibc
    down; the true value is a near-tie), so the point may sit on either side of
    the node -- weight 0 in one cell or 1 - 1e-16 in the next, the same template"""
    u, m = libdict['uvec0'], rag['interp/mapped'][i, 0]
    return bool(np.isfinite(m) and np.min(np.abs(u - m)) <= 4 * np.spacing(m))


def test_fixture_is_ragged(libdict, rag):
    lens = [len(libdict['uvec%d' % i]) for i in range(4)]
    assert lens == [9, 6, 5, 3] and libdict['idgrid'].shape == (9, 6, 5, 3)
    assert (libdict["idgrid"] < 0).sum() == 65     # 5 scattered holes + the 2x2x5x3 corner
    # every branch of GridInterp.__call__ is in the vectors
    assert all((rag['interp/branch'] == b).sum() >= 5 for b in (0, 1, 2))
    # cell indices that are legal in one dimension and beyond the end of
    # another (the case an equal-length grid cannot show)
    pos = rag['interp/pos'][rag['interp/branch'] == 0]
    assert pos[:, 0].max() >= 6 and pos[:, 1].max() >= 4


def test_oracle_polylinear_ragged(olib, libdict, rag):
    P = rag['interp/params']
    nsens = 0
    for i, p in enumerate(P):
        with np.errstate(all='ignore'):
            spec, info = olib.eval(p, details=True)
            o = olib.outside_flag(p)
        br = int(rag['interp/branch'][i])
        if libm_sensitive(libdict, rag, i):
            nsens += 1
            np.testing.assert_allclose(spec, rag['interp/eval'][i], rtol=3e-7)
            continue
        np.testing.assert_array_equal(info['pos'], rag['interp/pos'][i])
        if br == 0:
            assert info['nearest'] < 0
            np.testing.assert_array_equal(info['ids'], rag['interp/ids'][i])
            # (log10 of the two interpreters' libm may differ in the last bit:
            # 1 ulp of log10 teff is 1e-14 of a cell width)
            np.testing.assert_allclose(info['weights'], rag['interp/weights'][i],
                                       rtol=1e-12, atol=1e-15)
            np.testing.assert_allclose(spec, rag['interp/eval'][i], rtol=1e-12)
        else:
            assert info['nearest'] == rag['interp/nearest'][i]
            np.testing.assert_allclose(spec, rag['interp/eval'][i], rtol=2e-7)
        oref = rag['interp/outside'][i]
        if np.isfinite(oref):
            assert abs(o - oref) <= 1e-12 * max(1, abs(oref))
        else:
            assert not np.isfinite(o)
    assert 1 <= nsens <= 5


@pytest.mark.parametrize('use_c', [False, True])
def test_oracle_get_chisq_ragged(olib, rag, use_c):
    sds = _sds(rag, orc.SpecData)
    for v, p, vs, want in zip(rag['chisq/vel'], rag['chisq/param'],
                              rag['chisq/vsini'], rag['chisq/value']):
        rot = None if np.isnan(vs) else (float(vs), )
        with np.errstate(all='ignore'):
            val = orc.get_chisq(sds, float(v), p, rot, options=dict(npoly=10),
                                config=CONFIG, libs={SETUP: olib}, use_c=use_c)
        assert abs(val - want) <= 1e-7 * max(abs(want), 1e3), (p, val, want)


# ------------------------------------------------------------------ GPU ------
@pytest.fixture(scope='module')
def config(libdict):
    from rvspecfit_amd import _lib, spec_inter
    from rvspecfit_amd.library import TemplateLibrary
    _lib.require_gpu()
    spec_inter.register_library(TemplateLibrary(SETUP, libdict), 'ragged://')
    return dict(CONFIG, template_lib='ragged://')


@pytest.mark.gpu
@pytest.mark.parametrize('mapped', [True, False])
def test_polylinear_ragged(rag, libdict, config, mapped):
    """rvs_template_polylinear: branch, vertex ids, nearest index bit-exact;
    weights 1e-13; template 1e-12 (float32-exp rows bit for bit).  mapped=True
    feeds the reference's mapped parameters (no log10 on the way: every point
    exact); mapped=False the physical ones (points ON a log10-mapped node are
    compared by value, see libm_sensitive)"""
    import torch
    from rvspecfit_amd import spec_inter
    it = spec_inter.getInterpolator(SETUP, config)
    P = rag['interp/mapped' if mapped else 'interp/params']
    with np.errstate(all='ignore'):
        templ, outside, cell, wts = it.lib.eval_batch(
            torch.as_tensor(P).to('cuda'), details=True, mapped=mapped)
    templ, outside = templ.cpu().numpy(), outside.cpu().numpy()
    cell, wts = cell.cpu().numpy(), wts.cpu().numpy()
    exact32 = host_numpy_expf_is_published_algorithm()
    for i in range(len(P)):
        br = int(rag['interp/branch'][i])
        if not mapped and libm_sensitive(libdict, rag, i):
            np.testing.assert_allclose(templ[i], rag['interp/eval'][i],
                                       rtol=3e-7)
            continue
        assert cell[i, 0] == br, (i, P[i])
        ref = rag['interp/eval'][i]
        if br == 0:
            np.testing.assert_array_equal(cell[i, 2:], rag['interp/ids'][i])
            np.testing.assert_allclose(wts[i], rag['interp/weights'][i],
                                       rtol=1e-13 if mapped else 1e-12,
                                       atol=1e-16 if mapped else 1e-15)
            np.testing.assert_allclose(templ[i], ref, rtol=1e-12)
            assert outside[i] == 0
        else:
            assert cell[i, 1] == rag['interp/nearest'][i], (i, P[i])
            # the capture ran numpy 1.26's float32 exp; where this host's numpy
            # runs the same published algorithm the rows agree bit for bit
            if exact32:
                np.testing.assert_array_equal(templ[i], ref)
            else:
                np.testing.assert_allclose(templ[i], ref, rtol=3e-7)
        oref = rag['interp/outside'][i]
        if np.isfinite(oref):
            assert abs(outside[i] - oref) <= 1e-12 * max(1, abs(oref))
        else:
            assert not np.isfinite(outside[i])


@pytest.mark.gpu
def test_objective_cell_search_ragged(rag, config):
    """the 4 + 16-thread cell search inside rvs_objective_fused against the
    stand-alone kernel chain and the reference's get_chisq, in cells whose
    index differs per dimension, on holes and outside the grid"""
    import torch
    from rvspecfit_amd import engine, spec_fit
    sds = _sds(rag, spec_fit.SpecData)
    b, _ = spec_fit.as_batch(sds)
    vs_all = rag['chisq/vsini']
    for with_rot in (False, True):
        ii = np.nonzero(np.isfinite(vs_all) == with_rot)[0]
        vel = torch.as_tensor(rag['chisq/vel'][ii]).to('cuda')
        par = torch.as_tensor(rag['chisq/param'][ii]).to('cuda')
        vs = torch.as_tensor(vs_all[ii]).to('cuda') if with_rot else None
        idx = torch.zeros(len(ii), dtype=torch.long, device='cuda')
        out = {}
        for fused in (True, False):
            engine.FUSED_OBJECTIVE = fused
            try:
                with np.errstate(all='ignore'):
                    out[fused] = spec_fit.chisq_jobs(b, idx, vel, par, vs,
                                                     dict(npoly=10), config)
            finally:
                engine.FUSED_OBJECTIVE = True
        c1, s1 = out[True]
        c0, s0 = out[False]
        assert torch.equal(s0, s1)
        for k, i in enumerate(ii):
            want = float(rag['chisq/value'][i])
            sc = max(abs(want), 1e3)
            assert abs(c1[k].item() - c0[k].item()) < 1e-11 * sc, (i, k)
            assert abs(c1[k].item() - want) < 1e-6 * sc, (i, c1[k].item(), want)


@pytest.mark.gpu
def test_get_chisq_api_ragged(rag, config):
    from rvspecfit_amd import spec_fit
    sds = _sds(rag, spec_fit.SpecData)
    for v, p, vs, want in zip(rag['chisq/vel'], rag['chisq/param'],
                              rag['chisq/vsini'], rag['chisq/value']):
        rot = None if np.isnan(vs) else (float(vs), )
        with np.errstate(all='ignore'):
            val = spec_fit.get_chisq(sds, float(v), tuple(p), rot,
                                     options=dict(npoly=10), config=config)
        assert abs(val - want) <= 1e-7 * max(abs(want), 1e3), (p, val, want)


@pytest.mark.gpu
def test_plugin_ragged(tmp_path, rag, libdict, olib):
    """plugin.Evaluator / Outside (interpolation_type 'generic',
    spec_inter.py:371-378) fed the MAPPED parameters, as SpecInterpolator does"""
    from rvspecfit_amd import plugin
    np.savez(os.path.join(tmp_path, 'rvsgpu_%s.npz' % SETUP), **libdict)
    fd = plugin.record(SETUP, os.path.join(tmp_path, 'rvsgpu_%s.npz' % SETUP))
    fd['template_lib'] = str(tmp_path)
    ev, out = plugin.Evaluator(fd), plugin.Outside(fd)
    for i, mp in enumerate(rag['interp/mapped']):
        br = int(rag['interp/branch'][i])
        t = ev(mp)
        np.testing.assert_allclose(t, rag['interp/eval'][i],
                                   rtol=1e-12 if br == 0 else 3e-7)
        o = out(mp)
        oref = rag['interp/outside'][i]
        if np.isfinite(oref):
            assert abs(o - oref) <= 1e-12 * max(1, abs(oref))
        else:
            assert not np.isfinite(o)


@pytest.mark.gpu
def test_process_ragged(rag, config):
    """vel_fit.process (device Nelder-Mead + fused objective + Hessian) on the
    ragged library against the reference's run; tolerances of
    test_process_golden"""
    from rvspecfit_amd import spec_fit, vel_fit
    sds = _sds(rag, spec_fit.SpecData)
    g = rag
    pd0 = {str(k): float(v) for k, v in zip(g['process/start_keys'],
                                            g['process/start_vals'])}
    r = vel_fit.process(sds, pd0, fixParam=None, options=dict(npoly=10),
                        config=config, priors=None)
    t = 'process/'
    assert r['minimize_success'] == bool(g[t + 'minimize_success'])
    assert abs(r['vel'] - g[t + 'vel']) < 0.01
    assert abs(r['vel_err'] / g[t + 'vel_err'] - 1) < 1e-2
    assert abs(r['chisq'] - g[t + 'chisq']) < 2e-3
    assert abs(r['chisq'] / g[t + 'chisq'] - 1) < 1e-6
    names = ['teff', 'logg', 'feh', 'alpha']
    got = np.array([r['param'][_] for _ in names])
    err = g[t + 'param_err']
    assert np.all(np.abs(got - g[t + 'param']) < 0.02 * err + 1e-9)
    assert abs(r['vsini'] - g[t + 'vsini']) < 0.05
    assert r['npix_array'] == [int(_) for _ in g[t + 'npix_array']]
    assert r['bad_hessian'] == bool(g[t + 'bad_hessian'])
    gerr = np.array([r['param_err'][_] for _ in names])
    np.testing.assert_allclose(gerr, err, rtol=2e-2)
