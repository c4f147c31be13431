"""Parity of the HIP path (through the C-ABI) against the CPU oracle and the
golden vectors captured from the reference.  Needs a real MI355X.

Tolerances (north star): chi^2 within 1e-6 relative (we assert tighter where
the arithmetic allows), RV within 0.01 km/s, integer / index work bit-exact.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, GOLD_CONFIG, gold_specdata, gold_lib_dict
from oracle import rvs_oracle as orc
from test_numpy_expf import host_numpy_expf_is_published_algorithm
from rvspecfit_amd import _lib

pytestmark = pytest.mark.gpu
# (False only on a host whose numpy float32 exp is not the AVX2 / AVX-512 one)
NP_EXPF_OK = host_numpy_expf_is_published_algorithm()

TAGS = ['c0', 'c1', 'c2', 'c3']
CHI_RTOL = 1e-8      # well inside the 1e-6 contract
RV_ATOL = 1e-3       # km/s, contract is 1e-2


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))


@pytest.fixture(scope='module')
def gpu():
    from rvspecfit_amd import _lib
    _lib.require_gpu()
    _lib.lib()
    return torch.device('cuda')


@pytest.fixture(scope='module')
def config(gpu):
    from rvspecfit_amd import spec_inter
    from rvspecfit_amd.library import TemplateLibrary
    cfg = dict(GOLD_CONFIG)
    cfg['template_lib'] = 'golden://'
    for n in ('gold_b', 'gold_r'):
        lib = TemplateLibrary(n, gold_lib_dict(n))
        spec_inter.register_library(lib, 'golden://')
    return cfg


def test_native_library_is_loaded(gpu):
    from rvspecfit_amd import _lib
    assert _lib.lib().rvs_abi_version() == _lib.ABI_VERSION
    maps = open('/proc/self/maps').read()
    assert 'librvsgpu.so' in maps


@pytest.mark.parametrize('kind', ['log', 'lin'])
def test_spline_construct_eval(cases, gpu, kind):
    from rvspecfit_amd import spec_fit
    g = lambda k: cases['spline/%s/%s' % (kind, k)]
    S = spec_fit.getRVInterpol(g('xs'), g('ys'), log_step=(kind == 'log'))
    co = S.coef[0].cpu().numpy()
    for i, k in enumerate('ABCD'):
        np.testing.assert_allclose(co[:-1, i], g(k), rtol=1e-9, atol=1e-12)
    ret, pos = S(g('evalx'), return_pos=True)
    np.testing.assert_allclose(ret, g('ret'), rtol=1e-11, atol=1e-12)
    O = orc.Spline(g('xs'), g('ys'), log_step=(kind == 'log'))
    _, opos = O(g('evalx'), return_pos=True)
    np.testing.assert_array_equal(pos, opos)   # integer work: bit exact


def test_spline_desi_size_vs_oracle(gpu):
    from rvspecfit_amd import spec_fit
    rng = np.random.RandomState(5)
    xs = np.exp(np.linspace(np.log(3500.), np.log(5900.), 6215))
    ys = 1 + 0.3 * rng.standard_normal(len(xs))
    ex = np.sort(rng.uniform(3600, 5800, size=2751))
    S = spec_fit.getRVInterpol(xs, ys)
    O = orc.Spline(xs, ys)
    co = S.coef[0].cpu().numpy()
    for i, k in enumerate('ABCD'):
        np.testing.assert_allclose(co[:-1, i], getattr(O, k), rtol=1e-9,
                                   atol=1e-11)
    ret, pos = S(ex, return_pos=True)
    oret, opos = O(ex, return_pos=True)
    np.testing.assert_array_equal(pos, opos)
    np.testing.assert_allclose(ret, oret, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize('ntp', [4, 5, 37, 300, 2047, 2048, 2049, 6215, 9001])
def test_spline_construct_forms(gpu, ntp):
    """exact (chunk carries) and windowed (form | 2) solves, A,B,C,D and power
    form, against the oracle's restatement of spliner.c construct"""
    from rvspecfit_amd import _lib
    rng = np.random.RandomState(ntp)
    B = 3
    for log_step in (True, False):
        xs = np.linspace(np.log(3500.), np.log(5900.), ntp)
        xs = np.exp(xs) if log_step else np.linspace(3500., 5900., ntp)
        ys = 1 + 0.3 * rng.standard_normal((B, ntp))
        ys[1] = np.exp(-0.5 * ((xs - 4500.) / 30.)**2) * 1e3   # smooth, big
        kn = torch.as_tensor(xs).to('cuda')
        yt = torch.as_tensor(ys).to('cuda')
        out = {}
        fac = torch.empty(_lib.lib().rvs_spline_factors_len(ntp),
                          dtype=torch.float64, device='cuda')
        assert _lib.lib().rvs_spline_factors(_lib.ptr(kn), ntp, _lib.ptr(fac),
                                             _lib.stream()) == 0
        for form in (0, 1, 2, 3):
            coef = torch.empty((B, ntp, 4), dtype=torch.float64, device='cuda')
            rc = _lib.lib().rvs_spline_construct(_lib.ptr(kn), _lib.ptr(yt), ntp,
                                                 B, form,
                                                 _lib.ptr(fac) if form & 2
                                                 else None, _lib.ptr(coef),
                                                 _lib.stream())
            assert rc == 0
            out[form] = coef.cpu().numpy()
        for b in range(B):
            O = orc.Spline(xs, ys[b], log_step=log_step)
            sc = np.abs(ys[b]).max() / np.diff(xs).min()
            for form in (0, 2):
                for i, k in enumerate('ABCD'):
                    np.testing.assert_allclose(out[form][b, :-1, i],
                                               getattr(O, k), rtol=1e-9,
                                               atol=1e-11 * sc)
            # power form: same cubic -> same values at interior points
            h = np.diff(xs)
            for form in (1, 3):
                c = out[form][b, :-1]
                for fr in (0.0, 0.3, 0.99):
                    dl = fr * h
                    v = c[:, 0] + dl * (c[:, 1] + dl * (c[:, 2] + dl * c[:, 3]))
                    dr = h - dl
                    w = O.A * dl**3 + O.B * dr**3 + O.C * dl + O.D * dr
                    np.testing.assert_allclose(v, w, rtol=1e-10,
                                               atol=1e-11 * np.abs(ys[b]).max())
            np.testing.assert_allclose(out[3][b], out[1][b], rtol=1e-9,
                                       atol=1e-11 * sc)


def test_spline_error_codes(gpu):
    from rvspecfit_amd import spec_fit
    xs = np.exp(np.linspace(1, 2, 50))
    S = spec_fit.getRVInterpol(xs, np.ones(50))
    with pytest.raises(AssertionError):
        S(np.array([xs[0] * 0.9, xs[3]]))
    with pytest.raises(AssertionError):
        S(np.array([xs[3], xs[-1]]))
    xs2 = xs.copy()
    xs2[2] *= 1.001
    with pytest.raises(AssertionError):
        spec_fit.getRVInterpol(xs2, np.ones(50))(np.array([xs[5]]))


@pytest.mark.parametrize('name', ['gold_b', 'gold_r'])
def test_polylinear(cases, gold_libs, config, name):
    from rvspecfit_amd import spec_inter
    it = spec_inter.getInterpolator(name, config)
    P = cases['interp/params']
    templ, outside, cell, wts = it.lib.eval_batch(
        torch.as_tensor(P).to('cuda'), details=True)
    templ, outside = templ.cpu().numpy(), outside.cpu().numpy()
    cell, wts = cell.cpu().numpy(), wts.cpu().numpy()
    olib = gold_libs[name]
    for i, p in enumerate(P):
        ospec, info = olib.eval(p, details=True)
        if info['nearest'] < 0:   # polylinear cell: ids bit exact
            assert cell[i, 0] == 0
            np.testing.assert_array_equal(cell[i, 2:], info['ids'])
            np.testing.assert_allclose(wts[i], info['weights'], rtol=1e-13,
                                       atol=1e-16)
            np.testing.assert_allclose(templ[i], ospec, rtol=1e-12)
        else:
            assert cell[i, 0] in (1, 2)
            assert cell[i, 1] == info['nearest']
            # float32 np.exp of the reference (spec_inter.py:160) restated
            # operation for operation (csrc/common.h:np_expf): bit for bit,
            # against the oracle's np.exp on this host and against the
            # reference's captured output
            if NP_EXPF_OK:
                np.testing.assert_array_equal(templ[i], ospec)
                np.testing.assert_array_equal(
                    templ[i], cases['interp/%s/eval' % name][i])
            else:
                np.testing.assert_allclose(templ[i], ospec, rtol=3e-7)
        np.testing.assert_allclose(templ[i], cases['interp/%s/eval' % name][i],
                                   rtol=3e-7 if info['nearest'] >= 0 else 1e-12)
        oref = cases['interp/%s/outside' % name][i]
        if np.isfinite(oref):
            assert abs(outside[i] - oref) <= 1e-12 * max(1, abs(oref))
        else:
            assert not np.isfinite(outside[i])
    # API of the reference object
    np.testing.assert_allclose(it.eval(tuple(P[0])), templ[0])
    assert it.outsideFlag(tuple(P[4])) == outside[4]
    with pytest.raises(ValueError):
        it.eval(dict(teff=5000., logg=2.))


def test_vsini(cases, config, gold_libs):
    from rvspecfit_amd import spec_fit
    lam = gold_libs['gold_b'].lam
    for i, v in enumerate(cases['vsini/vsinis']):
        out = spec_fit.convolve_vsini(lam, cases['vsini/templ'], float(v))
        np.testing.assert_allclose(out, cases['vsini/conv_%d' % i], rtol=1e-11)
    P = cases['interp/params']
    rots = [None, (10., ), (300., )]
    for name in ('gold_b', 'gold_r'):
        for ip in (0, 3, 4):
            for ir, rot in enumerate(rots):
                o, lam_t, sp, tag, ls = spec_fit.getCurTempl(
                    name, tuple(P[ip]), rot, config)
                k = 'curtempl/%s/p%d_r%d/' % (name, ip, ir)
                np.testing.assert_allclose(sp, cases[k + 'spec'], rtol=3e-7)
                assert abs(o - cases[k + 'outside']) < 1e-12


def _sds(cases, tag):
    from rvspecfit_amd import spec_fit
    return gold_specdata(cases, tag, spec_fit.SpecData)


@pytest.mark.parametrize('tag', TAGS)
def test_get_chisq(cases, config, tag):
    from rvspecfit_amd import spec_fit
    sds = _sds(cases, tag)
    for i in range(7):
        k = '%s/chisq/t%d/' % (tag, i)
        vs = float(cases[k + 'vsini'])
        rot = None if np.isnan(vs) else (vs, )
        opt = dict(npoly=int(cases[k + 'npoly']),
                   rbf_continuum=bool(cases[k + 'rbf']))
        val = spec_fit.get_chisq(sds, float(cases[k + 'vel']),
                                 tuple(cases[k + 'param']), rot, options=opt,
                                 config=config)
        ref = float(cases[k + 'value'])
        # (trial 5 is a nearest-neighbour template, which the reference
        # exponentiates in float32, spec_inter.py:160; the device runs numpy's
        # float32 exp operation for operation)
        tol = 1e-7
        assert abs(val - ref) <= tol * abs(ref), (i, val, ref)
        if i < 3:
            full = spec_fit.get_chisq(sds, float(cases[k + 'vel']),
                                      tuple(cases[k + 'param']), rot,
                                      options=opt, config=config,
                                      full_output=True)
            np.testing.assert_allclose(full['chisq_array'],
                                       cases[k + 'chisq_array'], rtol=1e-6)
            np.testing.assert_array_equal(full['npix_array'],
                                          cases[k + 'npix_array'])
            for n, m, rm in zip(cases[tag + '/names'], full['models'],
                                full['raw_models']):
                np.testing.assert_allclose(m, cases[k + 'model_%s' % n],
                                           rtol=1e-6)
                np.testing.assert_allclose(rm, cases[k + 'raw_model_%s' % n],
                                           rtol=3e-7)
    val = spec_fit.get_chisq(sds, float(cases[tag + '/vel']),
                             tuple(cases[tag + '/truth']), None,
                             options=dict(npoly=10), config=config,
                             espec_systematic=0.05)
    assert abs(val - cases[tag + '/chisq/sys005']) <= 1e-7 * abs(val)


@pytest.mark.parametrize('tag', TAGS)
def test_find_best_vs_reference_and_oracle(cases, config, gold_libs,
                                           gold_config, tag):
    from rvspecfit_amd import spec_fit
    sds = _sds(cases, tag)
    osds = gold_specdata(cases, tag, orc.SpecData)
    vg = cases['vel_grid']
    for g in ('g1', 'g3'):
        k = '%s/%s/' % (tag, g)
        vs = float(cases[k + 'vsini'])
        rot = None if np.isnan(vs) else (vs, )
        pl = [tuple(_) for _ in cases[k + 'params_list']]
        r = spec_fit.find_best(sds, vg, pl, rot, options=dict(npoly=10),
                               config=config)
        assert abs(r['best_vel'] - cases[k + 'best_vel']) < RV_ATOL
        assert abs(r['vel_err'] - cases[k + 'vel_err']) < 1e-4
        assert abs(r['kurtosis'] - cases[k + 'kurtosis']) < 1e-4
        assert abs(r['skewness'] - cases[k + 'skewness']) < 1e-4
        np.testing.assert_allclose(r['best_param'], cases[k + 'best_param'])
        np.testing.assert_allclose(r['probs'], cases[k + 'probs'], rtol=1e-5,
                                   atol=1e-12)
        assert abs(r['best_chi'] - cases[k + 'best_chi']) <= 1e-7 * abs(
            r['best_chi'])
        # whole grid against the reference's grid
        b, _ = spec_fit.as_batch(sds)
        par = torch.as_tensor(np.array(pl))[None].to('cuda')
        vst = None if rot is None else torch.as_tensor([rot[0]],
                                                       dtype=torch.float64
                                                       ).to('cuda')
        chisq, st, _ = spec_fit.chisq_grid_jobs(
            b, torch.as_tensor(vg).to('cuda'), par, vst, dict(npoly=10), config)
        got = chisq[0].cpu().numpy().T   # [Nv, Np]
        assert rel(got, cases[k + 'chisq_grid']) < 1e-7
        assert int(st.sum().item()) == 0
        if g == 'g1':
            # and against the C oracle (tighter: same float32 template rows)
            fast = orc.chisq_grid_fast(osds, vg, pl[0], rot, dict(npoly=10),
                                       gold_config, gold_libs)
            assert rel(got[:, 0], fast) < CHI_RTOL


@pytest.mark.parametrize('tag', TAGS)
def test_chisq_continuum(cases, config, tag):
    from rvspecfit_amd import spec_fit
    r = spec_fit.get_chisq_continuum(_sds(cases, tag), options=dict(npoly=10))
    np.testing.assert_allclose(r['chisq_array'],
                               cases[tag + '/cont/chisq_array'], rtol=1e-8)
    np.testing.assert_allclose(r['redchisq_array'],
                               cases[tag + '/cont/redchisq_array'], rtol=1e-8)


@pytest.mark.parametrize('npoly,rbf', [(17, True), (24, False), (32, True)])
def test_npoly_above_16(cases, config, gold_config, gold_libs, npoly, rbf):
    """npoly 17 ... 32: no velocity-grid kernel (16 is its widest basis), but
    get_chisq(full_output), the point objective and get_chisq_continuum take them
    (rvs_chisq_full, FULL_MAXP 32; the continuum through rvs_chisq_full's tiers)
    and the basis tables for them are built on the device like any other"""
    from rvspecfit_amd import spec_fit
    tag = 'c1'
    sds = _sds(cases, tag)
    osds = gold_specdata(cases, tag, orc.SpecData)
    opt = dict(npoly=npoly, rbf_continuum=rbf)
    par = (5500., 3.0, -0.5, 0.2)
    got = spec_fit.get_chisq(sds, 23.0, par, rot_params=(20., ), options=opt,
                             config=config, full_output=True)
    want = orc.get_chisq(osds, 23.0, par, rot_params=(20., ), options=opt,
                         config=gold_config, libs=gold_libs, full_output=True)
    npix = sum(len(_.lam) for _ in osds)   # (-2 log L passes near 0: scale by npix)
    assert abs(got['chisq'] - want['chisq']) <= 1e-6 * max(abs(want['chisq']), npix)
    np.testing.assert_allclose(got['chisq_array'], want['chisq_array'], rtol=1e-6)
    for m, w in zip(got['models'], want['models']):
        assert np.abs(m - w).max() <= 1e-6 * np.abs(w).max()
    r = spec_fit.get_chisq_continuum(sds, options=opt)
    w = orc.get_chisq_continuum(osds, options=opt)
    np.testing.assert_allclose(r['chisq_array'], w['chisq_array'], rtol=1e-7)
    # ... and against the reference's own values (npoly_wide_cases.npz)
    g = np.load(os.path.join(GOLD, 'npoly_wide_cases.npz'))
    for tg in ('c0', 'c1'):
        sdg = _sds(cases, tg)
        k0 = '%s/p%d/' % (tg, npoly)
        assert bool(g[k0 + 'rbf']) == rbf
        rc = spec_fit.get_chisq_continuum(sdg, options=opt)
        np.testing.assert_allclose(rc['chisq_array'], g[k0 + 'cont/chisq_array'],
                                   rtol=1e-7)
        npg = sum(len(_.lam) for _ in sdg)
        for ip in range(2):
            k = k0 + 't%d/' % ip
            vs = float(g[k + 'vsini'])
            rot = None if np.isnan(vs) else (vs, )
            args = (sdg, float(g[k + 'vel']), tuple(g[k + 'param']))
            val = spec_fit.get_chisq(*args, rot_params=rot, options=opt, config=config)
            ref = float(g[k + 'value'])
            assert abs(val - ref) <= 1e-6 * max(abs(ref), npg), (tg, ip, val, ref)
            full = spec_fit.get_chisq(*args, rot_params=rot, options=opt,
                                      config=config, full_output=True)
            np.testing.assert_allclose(full['chisq_array'], g[k + 'chisq_array'],
                                       rtol=1e-6)
            np.testing.assert_array_equal(full['npix_array'], g[k + 'npix_array'])
            for n, m in zip(cases[tg + '/names'], full['models']):
                w_ = g[k + 'model_%s' % n]
                assert np.abs(m - w_).max() <= 1e-6 * np.abs(w_).max()


@pytest.mark.parametrize('tag,npoly', [('c0', 17), ('c1', 17), ('c0', 24), ('c1', 24)])
def test_velocity_grid_callers_above_16(cases, config, tag, npoly):
    """find_best and vel_fit.process at 17 / 24 continuum functions -- the reference
    has no cap (spec_fit.py:860, :1018-1092), the velocity-grid kernel keeps at most
    16 per lane: engine.chisq_grid sends every (job, velocity) of such a basis through
    rvs_chisq_full's arm values (the same route the ill-conditioned jobs take), the
    optimiser's objective likewise.  Against the reference's own find_best (whole
    chi^2 grid) and process runs (npoly_wide_grid_cases.npz,
    make_golden_npoly_wide.py)."""
    from rvspecfit_amd import spec_fit, vel_fit
    g = np.load(os.path.join(GOLD, 'npoly_wide_grid_cases.npz'))
    sds = _sds(cases, tag)
    k0 = '%s/p%d/' % (tag, npoly)
    opt = dict(npoly=npoly, rbf_continuum=bool(g[k0 + 'rbf']))
    vg = g[k0 + 'vel_grid']
    pl = [tuple(_) for _ in g[k0 + 'params']]
    npix = sum(len(_.lam) for _ in sds)
    for vt in ('rot', 'norot'):
        k = k0 + vt + '/'
        rot = (20., ) if vt == 'rot' else None
        r = spec_fit.find_best(sds, vg, pl, rot, options=opt, config=config)
        assert abs(r['best_vel'] - g[k + 'best_vel']) < RV_ATOL
        assert abs(r['vel_err'] - g[k + 'vel_err']) < 1e-4
        assert abs(r['kurtosis'] - g[k + 'kurtosis']) < 1e-4
        assert abs(r['skewness'] - g[k + 'skewness']) < 1e-4
        np.testing.assert_allclose(r['best_param'], g[k + 'best_param'])
        np.testing.assert_allclose(r['probs'], g[k + 'probs'], rtol=1e-5, atol=1e-12)
        assert abs(r['best_chi'] - g[k + 'best_chi']) <= 1e-6 * max(
            abs(float(g[k + 'best_chi'])), npix)
        b, _ = spec_fit.as_batch(sds)
        par = torch.as_tensor(np.array(pl))[None].to('cuda')
        vst = None if rot is None else torch.as_tensor(
            [rot[0]], dtype=torch.float64).to('cuda')
        chisq, st, _ = spec_fit.chisq_grid_jobs(
            b, torch.as_tensor(vg).to('cuda'), par, vst, opt, config)
        got = chisq[0].cpu().numpy().T   # [Nv, Np]
        # (-2 log L passes near zero: scaled by the pixel count)
        assert np.abs(got - g[k + 'chisq']).max() <= 1e-6 * max(
            np.abs(g[k + 'chisq']).max(), npix)
    # vel_fit.process end to end (tolerances of test_process_golden)
    kp = k0 + 'process/'
    pd0 = dict(zip([str(_) for _ in g[kp + 'start_keys']],
                   [float(_) for _ in g[kp + 'start_vals']]))
    cfg = dict(config, second_minimizer=False)
    r = vel_fit.process(sds, pd0, options=opt, config=cfg)
    assert r['minimize_success'] == bool(g[kp + 'minimize_success'])
    assert abs(r['vel'] - g[kp + 'vel']) < 0.01
    assert abs(r['vel_err'] / g[kp + 'vel_err'] - 1) < 1e-2
    assert abs(r['chisq'] - g[kp + 'chisq']) < 2e-3   # fatol-level
    names = ['teff', 'logg', 'feh', 'alpha']
    got = np.array([r['param'][_] for _ in names])
    err = g[kp + 'param_err']
    ok = np.isfinite(err) & (err > 0)
    assert np.all(np.abs(got - g[kp + 'param'])[ok] < 0.02 * err[ok] + 1e-9)
    np.testing.assert_allclose(r['chisq_array'], g[kp + 'chisq_array'], rtol=1e-5)
    # ... and a batch of such spectra takes the same route
    from rvspecfit_amd.engine import SpecBatch
    batch = SpecBatch.from_specdata([sds, sds])
    rb = vel_fit.process(batch, {k_: np.array([v_, v_]) for k_, v_ in pd0.items()},
                         options=opt, config=cfg)
    assert abs(float(rb['vel'][1]) - r['vel']) < 1e-9
    assert abs(float(rb['chisq'][0]) - r['chisq']) < 1e-9


def test_infinite_error_on_a_single_grid_is_data(cases, config):
    """espec = +inf marks the padding of a short grid in a grid set (G > 1) only.
    On an ordinary arm it is data: the reference takes log(inf) into the
    likelihood and raises (spec_fit.py:963-974)"""
    from rvspecfit_amd import spec_fit
    sds = _sds(cases, 'c1')
    es = np.array(sds[0].espec)
    es[100] = np.inf
    bad = [spec_fit.SpecData(sds[0].name, sds[0].lam, sds[0].spec, es,
                             badmask=sds[0].badmask)] + list(sds[1:])
    par = (5500., 3.0, -0.5, 0.2)
    with pytest.raises(RuntimeError):
        spec_fit.get_chisq(bad, 23.0, par, options=dict(npoly=10), config=config)
    with pytest.raises(RuntimeError):
        spec_fit.get_chisq(bad, 23.0, par, options=dict(npoly=10), config=config,
                           full_output=True)


@pytest.mark.parametrize('variant', ['quantised', 'plateau', 'inf_errors',
                                     'odd_even', 'negative'])
def test_ccf_preprocess_median_selection(cases, config, gold_libs, variant):
    """the medians of preprocess_data come from selection (key histograms over
    the block, bisection on the key bits per bin), not from sorts: inputs where
    that is delicate -- long runs of equal values, +inf errors inside
    nanmedian, odd and even counts, negative medians -- against the oracle"""
    from rvspecfit_amd import spec_fit, spec_inter, engine
    rng = np.random.RandomState(3)
    sds = _sds(cases, 'c1')
    osds = gold_specdata(cases, 'c1', orc.SpecData)
    new, onew = [], []
    for sd, osd in zip(sds, osds):
        spec, espec = np.array(osd.spec), np.array(osd.espec)
        bad = np.array(osd.badmask)
        n = len(spec)
        if variant == 'quantised':        # ~20 distinct flux values, 5 distinct errors
            q = np.median(spec) / 10
            spec = np.round(spec / q) * q
            espec = np.round(espec / np.median(espec) * 2 + 0.5) * np.median(espec) / 2
        elif variant == 'plateau':        # half the arm exactly constant
            spec[n // 4:3 * n // 4] = np.median(spec)
            espec[:] = np.median(espec)
        elif variant == 'inf_errors':     # nanmedian keeps +inf (then masked)
            espec[rng.choice(n, n // 3, replace=False)] = np.inf
        elif variant == 'odd_even':       # drop one pixel: the other parity of n
            spec, espec, bad = spec[:-1], espec[:-1], bad[:-1]
        elif variant == 'negative':       # sky-subtracted noise: negative medians
            spec = spec - 1.2 * np.median(spec)
        lam = np.array(osd.lam)[:len(spec)]
        new.append(spec_fit.SpecData(sd.name, lam, spec, espec, badmask=bad))
        onew.append(orc.SpecData(osd.name, lam, spec, espec, badmask=bad))
    b, _ = spec_fit.as_batch(new)
    libs = spec_inter.get_libs(b.names, config)
    for arm, osd in zip(b.arms, onew):
        if variant == 'odd_even' and arm.npix != len(osd.lam):
            pytest.skip('the CCF tables of the golden setup fix npix')
        pre = engine.ccf_preprocess(arm, libs[arm.name], config, details=True)
        ps, pi, info = orc.preprocess_data(osd.lam, osd.spec, osd.espec,
                                           gold_libs[osd.name].ccf,
                                           badmask=osd.badmask, details=True)
        gps = pre['proc_spec'][0].cpu().numpy()
        gpi = pre['proc_ivar'][0].cpu().numpy()
        np.testing.assert_array_equal(gpi == 0, pi == 0)
        np.testing.assert_array_equal(np.isfinite(gps), np.isfinite(ps))
        ok = np.isfinite(ps)
        np.testing.assert_allclose(pre['cont'][0].cpu().numpy(), info['cont'],
                                   rtol=5e-6)
        np.testing.assert_allclose(gps[ok], ps[ok], rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(gpi, pi, rtol=2e-5)


@pytest.mark.parametrize('tag', TAGS)
def test_ccf(cases, config, gold_libs, gold_config, tag):
    from rvspecfit_amd import fitter_ccf, spec_fit, spec_inter, engine
    sds = _sds(cases, tag)
    osds = gold_specdata(cases, tag, orc.SpecData)
    b, _ = spec_fit.as_batch(sds)
    libs = spec_inter.get_libs(b.names, config)
    for arm, osd in zip(b.arms, osds):
        pre = engine.ccf_preprocess(arm, libs[arm.name], config, details=True)
        ps, pi, info = orc.preprocess_data(osd.lam, osd.spec, osd.espec,
                                           gold_libs[osd.name].ccf,
                                           badmask=osd.badmask, details=True)
        gps, gpi = pre['proc_spec'][0].cpu().numpy(), pre['proc_ivar'][0].cpu().numpy()
        # masks are integer work: the zero pattern of ivar must be identical
        np.testing.assert_array_equal(gpi == 0, pi == 0)
        np.testing.assert_array_equal(gps == 0, ps == 0)
        # continuum: LM converges to the minimum TRF stops near (its 1e-8 tols)
        np.testing.assert_allclose(pre['cont'][0].cpu().numpy(), info['cont'],
                                   rtol=2e-6)
        np.testing.assert_allclose(gps, ps, rtol=5e-6, atol=1e-7)
        np.testing.assert_allclose(gpi, pi, rtol=5e-6)
        k = '%s/ccf/%s/' % (tag, arm.name)
        np.testing.assert_allclose(gps, cases[k + 'proc_spec'], rtol=5e-6,
                                   atol=1e-7)
    r = fitter_ccf.fit(sds, config)
    o = orc.ccf_fit(osds, gold_config, gold_libs)
    np.testing.assert_allclose([r['best_par'][k] for k in
                                ('teff', 'logg', 'feh', 'alpha')],
                               cases[tag + '/ccf/best_par'])
    assert abs(r['best_vel'] - cases[tag + '/ccf/best_vel']) < RV_ATOL
    assert abs(r['best_vel'] - o['best_vel']) < RV_ATOL
    np.testing.assert_allclose(r['best_ccf'], cases[tag + '/ccf/best_ccf'],
                               rtol=2e-6)
    np.testing.assert_allclose(r['vel_grid'], cases[tag + '/ccf/vel_grid'])
    bv = float(cases[tag + '/ccf/best_vsini'])
    assert r['best_vsini'] == bv
    for sd in sds:
        np.testing.assert_allclose(
            r['best_model'][sd.name],
            cases['%s/ccf/%s/best_model' % (tag, sd.name)])


@pytest.mark.parametrize('tag', TAGS)
def test_ccf_all_templates_vs_oracle(cases, config, gold_libs, gold_config,
                                     tag):
    """every (template, velocity) entry of the CCF chi^2 surface, computed
    from the SAME pre-processed spectrum as the oracle (isolates the FFT)."""
    from rvspecfit_amd import spec_fit, spec_inter, engine
    sds = _sds(cases, tag)
    osds = gold_specdata(cases, tag, orc.SpecData)
    b, _ = spec_fit.as_batch(sds)
    libs = spec_inter.get_libs(b.names, config)
    r = engine.ccf_fit(b, libs, config, keep_all=True)
    o = orc.ccf_fit(osds, gold_config, gold_libs, details=True)
    got = r['all_chisqs'][0].cpu().numpy()
    ref = o['all_chisqs']
    scale = np.abs(ref).max()
    assert np.max(np.abs(got - ref)) < 2e-5 * scale
    assert int(r['best_id'][0].item()) == o['best_id']


@pytest.mark.parametrize('tag', TAGS)
def test_ccf_nocontinuum(cases, config, gold_libs, gold_config, tag):
    """config['ccf_continuum_normalize'] = False selects the template set made
    by `rvs_make_ccf --nocontinuum` (fitter_ccf.py:40-47): no continuum fit and
    no error / negative-flux masks in preprocess_data (make_ccf.py:370-376),
    chi^2 = -c0^2 / c1 (fitter_ccf.py:204-207: two inverse transforms per
    template).  Against the reference's own run (nocont_cases.npz) and the
    oracle."""
    from rvspecfit_amd import fitter_ccf, spec_fit, spec_inter, engine
    nc = dict(np.load(os.path.join(GOLD, 'nocont_cases.npz')))
    cfg = dict(config, ccf_continuum_normalize=False)
    ocfg = dict(gold_config, ccf_continuum_normalize=False)
    sds = _sds(cases, tag)
    osds = gold_specdata(cases, tag, orc.SpecData)
    b, _ = spec_fit.as_batch(sds)
    libs = spec_inter.get_libs(b.names, cfg)
    for arm in b.arms:
        assert not libs[arm.name].ccf_set(cfg)['continuum']
        pre = engine.ccf_preprocess(arm, libs[arm.name], cfg)
        gps = pre['proc_spec'][0].cpu().numpy()
        gpi = pre['proc_ivar'][0].cpu().numpy()
        k = '%s/%s/' % (tag, arm.name)
        # masks are integer work: identical zero patterns
        np.testing.assert_array_equal(gpi == 0, nc[k + 'proc_ivar'] == 0)
        np.testing.assert_array_equal(gps == 0, nc[k + 'proc_spec'] == 0)
        np.testing.assert_allclose(gps, nc[k + 'proc_spec'], rtol=1e-10,
                                   atol=1e-11)
        np.testing.assert_allclose(gpi, nc[k + 'proc_ivar'], rtol=1e-10)
    r = engine.ccf_fit(b, libs, cfg, keep_all=True)
    o = orc.ccf_fit(osds, ocfg, gold_libs, details=True)
    got = r['all_chisqs'][0].cpu().numpy()
    scale = np.abs(o['all_chisqs']).max()
    assert np.max(np.abs(got - o['all_chisqs'])) < 2e-5 * scale
    np.testing.assert_allclose(got.min(axis=1), nc[tag + '/template_min'],
                               rtol=0, atol=2e-5 * scale)
    assert int(r['best_id'][0].item()) == int(nc[tag + '/best_id']) == o['best_id']
    f = fitter_ccf.fit(sds, cfg)
    np.testing.assert_allclose([f['best_par'][k] for k in
                                ('teff', 'logg', 'feh', 'alpha')],
                               nc[tag + '/best_par'])
    assert abs(f['best_vel'] - float(nc[tag + '/best_vel'])) < RV_ATOL
    np.testing.assert_allclose(f['best_ccf'], nc[tag + '/best_ccf'], rtol=0,
                               atol=2e-5 * scale)
    v = float(nc[tag + '/best_vsini'])
    assert (f['best_vsini'] is None and np.isnan(v)) or f['best_vsini'] == v
    for arm in b.arms:
        np.testing.assert_array_equal(
            f['best_model'][arm.name], nc['%s/%s/best_model' % (tag, arm.name)])
    # the default configuration still takes the continuum-normalised set
    d = fitter_ccf.fit(sds, config)
    np.testing.assert_allclose(d['best_ccf'], cases[tag + '/ccf/best_ccf'],
                               rtol=0, atol=2e-5 * np.abs(
                                   cases[tag + '/ccf/best_ccf']).max())


def test_ccf_set_missing_raises(gpu):
    """a setup converted without its --nocontinuum set: selecting it fails like
    the reference's missing ccf_nocont_<setup>.h5"""
    from rvspecfit_amd.library import TemplateLibrary
    lib = TemplateLibrary('gold_b', np.load(os.path.join(GOLD, 'lib_gold_b.npz')))
    assert lib.ccf_set({})['continuum']
    with pytest.raises(RuntimeError):
        lib.ccf_set(dict(ccf_continuum_normalize=False))


@pytest.mark.parametrize('tag', TAGS)
def test_refine(cases, config, tag):
    from rvspecfit_amd import vel_fit
    sds = _sds(cases, tag)
    bp = dict(params=tuple(cases[tag + '/truth']), rot_params=None)
    bv, be, sk, ku = vel_fit._find_best_vel_iterate(
        float(cases[tag + '/refine/start_vel']), config['min_vel'],
        config['max_vel'], config['vel_step0'], specdata=sds, best_param=bp,
        config=config, options=dict(npoly=10),
        min_vel_step=config['min_vel_step'])
    assert abs(bv - cases[tag + '/refine/best_vel']) < RV_ATOL
    assert abs(be - cases[tag + '/refine/vel_err']) < 1e-4
    assert abs(sk - cases[tag + '/refine/skewness']) < 1e-3
    assert abs(ku - cases[tag + '/refine/kurtosis']) < 1e-3


def test_firstguess(cases, config):
    from rvspecfit_amd import vel_fit
    sds = _sds(cases, 'c0')
    pg = {'logg': [1, 3], 'teff': [4000, 5000, 7000], 'feh': [-2, -1],
          'alpha': [0]}
    fg = vel_fit.firstguess(sds, options=dict(npoly=10), config=config,
                            vsinigrid=(None, 100), paramsgrid=pg)
    keys = [str(_) for _ in cases['c0/firstguess/keys']]
    assert sorted(fg.keys()) == keys
    np.testing.assert_allclose([float(fg[k]) for k in keys],
                               cases['c0/firstguess/vals'])


def test_find_best_one_list_shares_templates(cases, config):
    """find_best over one parameter list for a whole batch (the reference's
    call; vel_fit.firstguess: 60 parameter sets x 3 rotations) builds the Np
    templates once instead of once per spectrum: chi^2 grid, best parameters
    and moments are those of the per-spectrum form bit for bit"""
    from rvspecfit_amd import spec_fit
    from rvspecfit_amd.engine import SpecBatch
    S = 7
    lists = [_sds(cases, 'c1') for _ in range(S)]
    batch = SpecBatch.from_specdata(lists)
    rng = np.random.RandomState(3)
    for a in batch.arms:   # make the spectra differ
        a.spec.mul_(torch.as_tensor(
            1 + 0.05 * rng.normal(size=tuple(a.spec.shape))).to(a.spec.device))
    pl = [tuple(_) for _ in cases['c1/g3/params_list']] + [(3400., 2., -1., 0.2)]
    vg = cases['vel_grid'].astype(np.float64)
    for rot in (None, (30., )):
        a = spec_fit.find_best(batch, vg, pl, rot_params=rot,
                               options=dict(npoly=10), config=config)
        per = torch.as_tensor(np.array(pl))[None].expand(S, -1, -1).to('cuda')
        b = spec_fit.find_best(batch, vg, per.contiguous(), rot_params=rot,
                               options=dict(npoly=10), config=config)
        for k in ('chisq', 'best_chi', 'best_vel', 'vel_err', 'best_param',
                  'probs', 'status', 'i1', 'i2'):
            assert torch.equal(a[k], b[k]), (rot, k)


def test_batch_equals_singles(cases, config):
    """4 golden spectra of the 2-arm cases stacked in one batch give the same
    records as one-by-one calls (batching is arithmetic-neutral)."""
    from rvspecfit_amd import spec_fit, pipeline
    from rvspecfit_amd.engine import SpecBatch
    lists = [_sds(cases, t) for t in ('c1', 'c3')]
    batch = SpecBatch.from_specdata(lists)
    rec = pipeline.fit_batch(batch, config, options=dict(npoly=10)).cpu().numpy()
    for i, sl in enumerate(lists):
        one = pipeline.fit_batch(SpecBatch.from_specdata([sl]), config,
                                 options=dict(npoly=10)).cpu().numpy()
        np.testing.assert_array_equal(rec[i], one[0])


@pytest.mark.parametrize('S,npoly,npix', [(131, 5, 777), (64, 10, 777),
                                          (3, 16, 777), (5, 3, 37), (2, 10, 64),
                                          (2, 7, 129)])
def test_chisq_continuum_batched(S, npoly, npix):
    """rvs_chisq_continuum (one wave per spectrum, lanes over pixels) against the
    oracle's get_chisq_continuum on ragged batch sizes with masked pixels; rows
    shorter than a wave, exactly a wave, one pixel more than two waves"""
    from rvspecfit_amd import engine, synth
    from rvspecfit_amd.engine import ArmData, SpecBatch
    rng = np.random.RandomState(5 + S)
    lam = np.linspace(4000., 5000., npix)
    spec = 1 + 0.3 * rng.normal(size=(S, npix)) + np.linspace(0, 1, npix)
    espec = rng.uniform(0.1, 0.5, size=(S, npix))
    bad = rng.uniform(size=(S, npix)) < 0.1
    batch = SpecBatch([ArmData('x', lam, spec, espec, bad, device='cuda')])
    res = engine.chisq_continuum_fix(
        batch, engine.chisq_continuum(batch, npoly=npoly), npoly=npoly)[0]
    for i in range(S):
        sd = [orc.SpecData('x', lam, spec[i], espec[i], badmask=bad[i])]
        ref = orc.get_chisq_continuum(sd, options=dict(npoly=npoly))
        np.testing.assert_allclose(res['true_chisq'][i].item(),
                                   ref['chisq_array'][0], rtol=1e-8)
        assert res['ngood'][i].item() == int((~bad[i]).sum())


def test_overlap_error(cases, config):
    from rvspecfit_amd import spec_fit
    sds = _sds(cases, 'c0')
    bad = [spec_fit.SpecData('gold_b', sds[0].lam * 1.2, sds[0].spec,
                             sds[0].espec)]
    with pytest.raises(RuntimeError):
        spec_fit.get_chisq(bad, 0., tuple(cases['c0/truth']),
                           options=dict(npoly=10), config=config)


def test_chisq_grid_every_npoly_vs_objective_kernel(cases, config):
    """The velocity-grid kernel holds a basis row in SGPR tuples that cover its
    P doubles exactly (16/8/4/2-dword scalar loads, csrc/chisq.hip CgRow): every
    P from 1 to 16 is a different set of pieces.  chi^2 on a 70-point grid (one
    full wave + 6 packed left-over lanes, i.e. both kernel variants) against the
    optimiser's one-kernel objective (explicit residual norm, its own basis
    handling) at the same velocities."""
    from rvspecfit_amd import spec_fit, engine
    sds = _sds(cases, 'c1')
    b, _ = spec_fit.as_batch(sds)
    p = np.array(cases['c1/truth'], dtype=np.float64)
    vg = np.linspace(-310.0, 295.0, 70)
    par = torch.as_tensor(p[None, None, :]).to('cuda')
    keep = engine.CG_PACK_MIN_JOBS
    engine.CG_PACK_MIN_JOBS = 1
    try:
        for npoly in range(1, 17):
            g, st, _ = spec_fit.chisq_grid_jobs(
                b, torch.as_tensor(vg).to('cuda'), par, None,
                dict(npoly=npoly), config)
            assert int(st.sum().item()) == 0, npoly
            g = g.cpu().numpy().reshape(-1)
            idx = torch.zeros(len(vg), dtype=torch.long, device='cuda')
            c, st2 = spec_fit.chisq_jobs(
                b, idx, torch.as_tensor(vg).to('cuda'),
                torch.as_tensor(np.tile(p, (len(vg), 1))).to('cuda'), None,
                dict(npoly=npoly), config)
            c = c.cpu().numpy()
            assert np.all(np.isfinite(g)) and np.all(np.isfinite(c)), npoly
            # the grid kernel's D.D - y.y against the explicit residual: 1e-16 *
            # sum (s/e)^2 of rounding noise
            np.testing.assert_allclose(g, c, rtol=1e-9, atol=1e-7,
                                       err_msg='npoly %d' % npoly)
    finally:
        engine.CG_PACK_MIN_JOBS = keep


_lin_cfg = {}


def _linear_grid_config():
    """gold_b on a LINEAR wavelength grid, registered under its own template_lib"""
    if not _lin_cfg:
        from rvspecfit_amd import spec_inter
        from rvspecfit_amd.library import TemplateLibrary
        d = gold_lib_dict('gold_b')
        lam_log = np.asarray(d['lam'], dtype=np.float64)
        d['lam'] = np.linspace(lam_log[0], lam_log[-1], len(lam_log))
        d['log_step'] = np.array(False)
        lib = TemplateLibrary('gold_b', d)
        assert not lib.log_step
        spec_inter.register_library(lib, 'golden-lin://')
        _lin_cfg['cfg'] = dict(GOLD_CONFIG, template_lib='golden-lin://')
        _lin_cfg['d'] = d
    return _lin_cfg['cfg'], _lin_cfg['d']


def test_chisq_grid_linear_template_grid(cases, gpu):
    """A template library on a LINEAR wavelength grid (log_step False: the knot
    index is (int)((x - x0)/step), spliner.c:92-96): the velocity-grid kernel's
    linear-knot loop instance, full and packed waves, against the oracle's
    get_chisq and against the optimiser's one-kernel objective."""
    from rvspecfit_amd import spec_fit, engine
    cfg, d = _linear_grid_config()
    sd = [x for x in _sds(cases, 'c1') if x.name == 'gold_b'][:1]
    assert sd
    b, _ = spec_fit.as_batch(sd)
    p = np.array(cases['c1/truth'], dtype=np.float64)
    vg = np.linspace(-310.0, 295.0, 70)
    keep = engine.CG_PACK_MIN_JOBS
    engine.CG_PACK_MIN_JOBS = 1
    try:
        g, st, _ = spec_fit.chisq_grid_jobs(
            b, torch.as_tensor(vg).to('cuda'),
            torch.as_tensor(p[None, None, :]).to('cuda'), None, dict(npoly=10),
            cfg)
    finally:
        engine.CG_PACK_MIN_JOBS = keep
    assert int(st.sum().item()) == 0
    g = g.cpu().numpy().reshape(-1)
    idx = torch.zeros(len(vg), dtype=torch.long, device='cuda')
    c, _ = spec_fit.chisq_jobs(
        b, idx, torch.as_tensor(vg).to('cuda'),
        torch.as_tensor(np.tile(p, (len(vg), 1))).to('cuda'), None,
        dict(npoly=10), cfg)
    np.testing.assert_allclose(g, c.cpu().numpy(), rtol=1e-9, atol=1e-7)
    olib = {'gold_b': orc.Library(d)}
    osd = [orc.SpecData(x.name, x.lam, x.spec, x.espec, badmask=x.badmask)
           for x in sd]
    for i in (0, 33, 69):
        want = orc.get_chisq(osd, float(vg[i]), tuple(p), None,
                             options=dict(npoly=10), config=dict(GOLD_CONFIG),
                             libs=olib)
        assert abs(g[i] - want) < 1e-8 * max(abs(want), 1e3), (i, g[i], want)


@pytest.mark.parametrize('npix', [2, 3, 4, 5, 8])
def test_chisq_grid_short_arms(cases, config, npix):
    """The grid kernel's pixel loop runs one pixel ahead of itself (row k+1,
    coordinates k+2, clamped to the last pixel): arms of 2..8 pixels."""
    from rvspecfit_amd import spec_fit, engine
    sd0 = _sds(cases, 'c1')[0]
    i0 = len(sd0.lam) // 3
    sds = [spec_fit.SpecData(sd0.name, sd0.lam[i0:i0 + npix].copy(),
                             sd0.spec[i0:i0 + npix].copy(),
                             sd0.espec[i0:i0 + npix].copy())]
    b, _ = spec_fit.as_batch(sds)
    p = np.array(cases['c1/truth'], dtype=np.float64)
    vg = np.linspace(-310.0, 295.0, 70)
    keep = engine.CG_PACK_MIN_JOBS
    engine.CG_PACK_MIN_JOBS = 1
    try:
        g, st, _ = spec_fit.chisq_grid_jobs(
            b, torch.as_tensor(vg).to('cuda'),
            torch.as_tensor(p[None, None, :]).to('cuda'), None,
            dict(npoly=1), config)
    finally:
        engine.CG_PACK_MIN_JOBS = keep
    assert int(st.sum().item()) == 0
    idx = torch.zeros(len(vg), dtype=torch.long, device='cuda')
    c, _ = spec_fit.chisq_jobs(
        b, idx, torch.as_tensor(vg).to('cuda'),
        torch.as_tensor(np.tile(p, (len(vg), 1))).to('cuda'), None,
        dict(npoly=1), config)
    np.testing.assert_allclose(g.cpu().numpy().reshape(-1), c.cpu().numpy(),
                               rtol=1e-9, atol=1e-7)


def test_chisq_grid_wave_placement_bit_identical(cases, config):
    """A velocity's chi^2 does not depend on where the kernel computes it: in a
    wave of 64 velocities of one spectrum, or in a wave that packs the left-over
    velocities (Nv % 64) of several jobs (chisq_grid_kernel<P, TAIL>)."""
    from rvspecfit_amd import spec_fit
    sds = _sds(cases, 'c1')
    vg = cases['vel_grid'].astype(np.float64)          # 400 velocities: 6 x 64 + 16
    assert len(vg) % 64 == 16
    pl = [tuple(_) for _ in cases['c1/g3/params_list']]
    b, _ = spec_fit.as_batch(sds)
    par = torch.as_tensor(np.array(pl))[None].to('cuda')

    from rvspecfit_amd import engine

    def grid(v, npoly, pack):
        keep = engine.CG_PACK_MIN_JOBS
        engine.CG_PACK_MIN_JOBS = pack
        try:
            chisq, st, _ = spec_fit.chisq_grid_jobs(
                b, torch.as_tensor(np.ascontiguousarray(v)).to('cuda'), par,
                None, dict(npoly=npoly), config)
        finally:
            engine.CG_PACK_MIN_JOBS = keep
        assert int(st.sum().item()) == 0
        return chisq.cpu().numpy().reshape(-1, len(v))
    for npoly in (10, 7, 15):
        ragged = grid(vg, npoly, -1)     # 7 waves per job, the last with 16 lanes
        packed = grid(vg, npoly, 1)      # 6 waves per job + 4 jobs per packed wave
        np.testing.assert_array_equal(packed, ragged)
        # the last 64 as a grid of their own: all in one full wave
        np.testing.assert_array_equal(grid(vg[-64:], npoly, 1), ragged[:, -64:])
        # 24 left-over velocities: 2 jobs per packed wave
        np.testing.assert_array_equal(grid(vg[-24:], npoly, 1), ragged[:, -24:])
        # 5 left-over velocities: 12 jobs per packed wave
        np.testing.assert_array_equal(grid(vg[:69], npoly, 1), ragged[:, :69])
        # left-overs that do not divide 64 (flat (job, velocity) packing: a
        # job's velocities may straddle two waves): 40 of 40, the 36 left over
        # by a 100-point grid of the refinement loop; 61 of a 125-point grid
        # stay a ragged wave (more than 40)
        np.testing.assert_array_equal(grid(vg[-40:], npoly, 1), ragged[:, -40:])
        np.testing.assert_array_equal(grid(vg[:100], npoly, 1), ragged[:, :100])
        np.testing.assert_array_equal(grid(vg[-125:], npoly, 1),
                                      ragged[:, -125:])
    # The packed launch runs on a library-owned side stream per host thread:
    # two host threads, each on a torch stream of its own, at the same time
    import threading
    want = {k: grid(vg + 0.37 * k, 10, 1) for k in range(2)}
    got, errs = {}, []

    def run(k):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(3):
                    got[k] = grid(vg + 0.37 * k, 10, 1)
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
    keep_pack = engine.CG_PACK_MIN_JOBS
    th = [threading.Thread(target=run, args=(k, )) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    engine.CG_PACK_MIN_JOBS = keep_pack   # (grid() saves / restores it per call)
    assert not errs, errs
    for k in range(2):
        np.testing.assert_array_equal(got[k], want[k])


def _nn_lib(d, lam):
    from rvspecfit_amd.library import TemplateLibrary
    dd = dict(lam=lam, log_step=np.array(True), log_ids=np.array([0]),
              parnames=np.array(['teff', 'logg', 'feh', 'alpha']),
              nn_dims=d['dims'], nn_M=d['M'], nn_S=d['S'])
    if 'pts' in d:
        dd['nn_pts'] = d['pts']
    for i in range(len(d['dims']) - 1):
        dd['nn_W%d' % i] = d['W%d' % i]
        dd['nn_b%d' % i] = d['b%d' % i]
    return TemplateLibrary('nn_test', dd)


def test_nn_template_vs_reference_golden(gpu):
    d = dict(np.load(os.path.join(GOLD, 'nn_case.npz')))
    lam = np.exp(np.linspace(np.log(4000.), np.log(4100.), int(d['dims'][-1])))
    lib = _nn_lib(d, lam)
    templ, outside = lib.eval_batch(torch.as_tensor(d['params']).to('cuda'))
    # float32 MLP: the MFMA k-order differs from torch-CPU's; see DESIGN.md
    np.testing.assert_allclose(templ.cpu().numpy(), d['out'], rtol=3e-6)
    np.testing.assert_allclose(outside.cpu().numpy(), d['outside'], rtol=1e-5,
                               atol=1e-9)


def test_nn_outside_kernel_vs_numpy(gpu):
    """rvs_nn_outside through the C-ABI: OutsideInterpolator.__call__
    (nn/RVSInterpolator.py:63-71) restated in numpy on the Mapper's point, for
    points inside, outside and with a NaN coordinate; mapped and unmapped entry."""
    import ctypes
    import scipy.spatial
    from rvspecfit_amd import _lib
    d = dict(np.load(os.path.join(GOLD, 'nn_case.npz')))
    pts = np.asarray(d['pts'], dtype=np.float64)
    xe = np.ascontiguousarray(scipy.spatial.ConvexHull(pts[:, :2]).equations)
    ye = np.ascontiguousarray(scipy.spatial.ConvexHull(pts[:, 2:]).equations)
    rng = np.random.default_rng(7)
    B = 300
    par = np.column_stack([rng.uniform(2500, 13000, B), rng.uniform(-1, 6, B),
                           rng.uniform(-3, 1, B), rng.uniform(-0.5, 1.5, B)])
    par[:len(d['params'])] = d['params']
    par[B - 1, 2] = np.nan
    y = par.astype(np.float32)
    y[:, 0] = np.log10(y[:, 0].astype(np.float64)).astype(np.float32)
    mp = (y.astype(np.float64) - d['M']) / d['S']
    dx = (mp[:, :2] @ xe[:, :-1].T + xe[:, -1]).max(axis=1)
    dy = (mp[:, 2:] @ ye[:, :-1].T + ye[:, -1]).max(axis=1)
    with np.errstate(invalid='ignore'):
        want = np.maximum(np.maximum(dx, dy), 0)**2
    assert np.isnan(want[B - 1]) and (want == 0).sum() > 20 and (want > 0).sum() > 20
    L = _lib.lib()
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to('cuda')  # noqa: E731
    out = torch.empty(B, dtype=torch.float64, device='cuda')
    tM, tS, txe, tye = dev(d['M'].astype(np.float64)), dev(
        d['S'].astype(np.float64)), dev(xe), dev(ye)
    for mapped, p in ((0, par), (1, mp)):
        tp = dev(p)
        rc = L.rvs_nn_outside(_lib.ptr(tp), B, 4, 1, _lib.ptr(tM), _lib.ptr(tS),
                              mapped, _lib.ptr(txe), xe.shape[0], _lib.ptr(tye),
                              ye.shape[0], _lib.ptr(out), _lib.stream())
        assert rc == 0
        got = out.cpu().numpy()
        assert np.isnan(got[B - 1])
        ok = np.isfinite(want)
        np.testing.assert_allclose(got[ok], want[ok], rtol=1e-12, atol=1e-14)
        np.testing.assert_array_equal(got[ok] == 0, want[ok] == 0)
    # the golden points: the reference's own values
    np.testing.assert_allclose(got[:len(d['outside'])], d['outside'], rtol=1e-5,
                               atol=1e-9)
    assert L.rvs_nn_outside(_lib.ptr(tp), 0, 4, 1, _lib.ptr(tM), _lib.ptr(tS), 0,
                            _lib.ptr(txe), xe.shape[0], _lib.ptr(tye),
                            ye.shape[0], _lib.ptr(out), None) < 0


def test_nn_template_desi_size_vs_oracle(gpu):
    """the production shape 4 -> 256 -> 256 -> 256 -> 200 -> 6215 on the f32
    MFMA path against the numpy float32 oracle, 300 parameter vectors (covers
    partial tiles in every dimension)"""
    rng = np.random.RandomState(12)
    dims = np.array([4, 256, 256, 256, 200, 6215], dtype=np.int32)
    d = dict(dims=dims, M=np.array([3.7, 2.5, -1., 0.5]),
             S=np.array([0.15, 1.4, 0.6, 0.3]))
    for i in range(5):
        k, n = dims[i], dims[i + 1]
        d['W%d' % i] = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
        d['b%d' % i] = (0.1 * rng.standard_normal(n)).astype(np.float32)
    lam = np.exp(np.linspace(np.log(3500.), np.log(5900.), 6215))
    lib = _nn_lib(d, lam)
    P = np.array([rng.uniform(3500, 9000, 300), rng.uniform(0, 5, 300),
                  rng.uniform(-2, 0, 300), rng.uniform(0, 1, 300)]).T
    templ, _ = lib.eval_batch(torch.as_tensor(P).to('cuda'))
    W = [(d['W%d' % i], d['b%d' % i]) for i in range(5)]
    ref = orc.nn_forward(W, P, d['M'], d['S'])
    np.testing.assert_allclose(templ.cpu().numpy(), ref, rtol=5e-6)


@pytest.mark.parametrize('B,npc,npix', [(3000, 200, 6215), (2700, 200, 5303),
                                        (10000, 200, 6449), (2800, 128, 6215),
                                        (2700, 256, 6200), (4000, 200, 6100)])
def test_nn_wide_layer_pipelined_equals_generic(gpu, B, npc, npix):
    """the wide last layer through nn_final_pipe_kernel (epilogue of a tile under the
    next tile's matrix products, XCD-owned row tiles, buffer stores that drop what
    lies outside the matrix) against nn_linear_kernel (option nn_pipe = 0): same MFMA
    order per output, same float64 exp -- BIT FOR BIT, on row counts that end in
    partial tiles, output widths with and without a partial column tile, the three
    inner widths the kernel is built for"""
    rng = np.random.RandomState(21)
    dims = np.array([4, 256, 256, 256, npc, npix], dtype=np.int32)
    d = dict(dims=dims, M=np.array([3.7, 2.5, -1., 0.5]),
             S=np.array([0.15, 1.4, 0.6, 0.3]))
    for i in range(5):
        k, n = dims[i], dims[i + 1]
        d['W%d' % i] = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
        d['b%d' % i] = (0.1 * rng.standard_normal(n)).astype(np.float32)
    lib = _nn_lib(d, np.exp(np.linspace(np.log(3500.), np.log(5900.), npix)))
    P = torch.as_tensor(np.array([rng.uniform(3500, 9000, B), rng.uniform(0, 5, B),
                                  rng.uniform(-2, 0, B), rng.uniform(0, 1, B)]).T
                        ).to('cuda').contiguous()
    # the rows behind the result are a guard band: a partial row tile's stores
    # behind row B must go nowhere
    big = torch.full((B + 200, npix), -7.0, dtype=torch.float64, device='cuda')
    out = torch.empty(B, dtype=torch.float64, device='cuda')
    got, _ = lib._eval_nn(P, big[:B], out)
    assert bool((big[B:] == -7.0).all())
    with _lib.option('nn_pipe', 0):
        want, _ = lib.eval_batch(P)
    assert torch.equal(got, want)
    assert torch.isfinite(got).all() and float(got.min()) > 0


@pytest.mark.parametrize('B', [7, 300, 700, 3000])
@pytest.mark.parametrize('same_shape', [True, False])
def test_nn_template_arms_equals_per_arm_calls(gpu, B, same_shape):
    """rvs_template_nn_arms (the arms' MLPs in grouped launches, grid.y = arm)
    gives the rows and outside flags of the per-arm rvs_template_nn /
    rvs_nn_outside calls BIT FOR BIT; arms whose networks differ in shape fall
    back to the per-arm calls inside the entry point."""
    import ctypes
    import scipy.spatial
    from rvspecfit_amd import _lib
    rng = np.random.RandomState(5)
    widths = [6215, 5303, 6449]
    libs = []
    for a, w in enumerate(widths):
        hid = (256, 256, 256, 200) if (same_shape or a != 1) else (256, 128, 200)
        dims = np.array((4, ) + hid + (w, ), dtype=np.int32)
        d = dict(dims=dims, M=np.array([3.7, 2.5, -1., 0.5]) + 0.01 * a,
                 S=np.array([0.15, 1.4, 0.6, 0.3]))
        for i in range(len(dims) - 1):
            k, n = dims[i], dims[i + 1]
            d['W%d' % i] = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(
                np.float32)
            d['b%d' % i] = (0.1 * rng.standard_normal(n)).astype(np.float32)
        if a != 2:   # the third arm has no hull: outside = 0
            d['pts'] = rng.uniform(-1.5, 1.5, (60, 4))
        libs.append(_nn_lib(d, np.exp(np.linspace(8.2, 8.7, w))))
    P = np.array([rng.uniform(3500, 9000, B), rng.uniform(0, 5, B),
                  rng.uniform(-2, 0, B), rng.uniform(0, 1, B)]).T
    tp = torch.as_tensor(np.ascontiguousarray(P)).to('cuda')
    want = [lib.eval_batch(tp) for lib in libs]
    arr = (_lib.NmNNArm * 3)()
    keep, got = [], []
    for a, lib in enumerate(libs):
        nl = len(lib.nn_W)
        Wp = (ctypes.c_void_p * nl)(*[w.data_ptr() for w in lib.nn_W])
        bp = (ctypes.c_void_p * nl)(*[x.data_ptr() for x in lib.nn_b])
        a0 = torch.empty((B, lib.nn_width()), dtype=torch.float32, device='cuda')
        a1 = torch.empty_like(a0)
        t = torch.empty((B, lib.ntp), dtype=torch.float64, device='cuda')
        o = torch.full((B, ), -1.0, dtype=torch.float64, device='cuda')
        keep += [Wp, bp, a0, a1]
        got.append((t, o))
        x = arr[a]
        x.M, x.S = lib.nn_M.data_ptr(), lib.nn_S.data_ptr()
        x.W, x.b = ctypes.cast(Wp, ctypes.c_void_p), ctypes.cast(bp, ctypes.c_void_p)
        x.dims = lib.nn_dims.ctypes.data
        x.act0, x.act1 = a0.data_ptr(), a1.data_ptr()
        x.templ, x.outside = t.data_ptr(), o.data_ptr()
        hull = lib.hull_device()
        if hull is not None:
            x.xeqs, x.yeqs = hull[0].data_ptr(), hull[1].data_ptr()
            x.nfx, x.nfy = hull[0].shape[0], hull[1].shape[0]
        x.nlayer, x.log_mask = nl, lib.log_mask
    rc = _lib.lib().rvs_template_nn_arms(_lib.ptr(tp), B, 4, 3,
                                         ctypes.addressof(arr), _lib.stream())
    assert rc == 0
    for (t, o), (wt, wo) in zip(got, want):
        assert torch.equal(t, wt)
        assert torch.equal(o, wo)
    assert float(got[2][1].abs().max()) == 0.0 and float(got[0][1].max()) > 0
    if not same_shape:
        return
    # with a row count on the device (what the lock-step optimiser passes): the rows
    # before it as above, the rows behind it untouched -- in the hidden stack, the
    # outside flags and the wide layer (pipelined kernel at 3000 rows)
    for n in sorted({0, 1, B // 3, B - 1, B}):
        for t, o in got:
            t.fill_(-5.0)
            o.fill_(-5.0)
        cnt = torch.tensor([n], dtype=torch.int32, device='cuda')
        rc = _lib.lib().rvs_template_nn_arms_n(_lib.ptr(tp), B, _lib.ptr(cnt), 4, 3,
                                               ctypes.addressof(arr), _lib.stream())
        assert rc == 0
        for (t, o), (wt, wo) in zip(got, want):
            assert torch.equal(t[:n], wt[:n]) and torch.equal(o[:n], wo[:n])
            assert bool((t[n:] == -5.0).all()) and bool((o[n:] == -5.0).all())


@pytest.mark.parametrize('dims', [
    (4, 96, 128, 50, 777),        # fused narrow layers, partial column tiles
    (4, 64, 333),                 # one hidden layer
    (4, 100, 72, 36, 515),        # widths not multiples of 32: layer by layer
    (4, 320, 64, 1000),           # a hidden layer wider than 256: layer by layer
    (3, 32, 32, 32, 32, 32, 32, 32, 32, 90),   # more layers than the fused kernel holds
])
def test_nn_template_layer_shapes_vs_oracle(gpu, dims):
    """rvs_template_nn picks between the one-launch hidden stack and the
    layer-by-layer kernels by the layer widths; both against the numpy oracle,
    with a row count that leaves partial row tiles"""
    rng = np.random.RandomState(len(dims) * 1000 + dims[1])
    dims = np.array(dims, dtype=np.int32)
    nd = int(dims[0])
    d = dict(dims=dims, M=np.array([3.7, 2.5, -1., 0.5])[:nd],
             S=np.array([0.15, 1.4, 0.6, 0.3])[:nd])
    nl = len(dims) - 1
    for i in range(nl):
        k, n = dims[i], dims[i + 1]
        d['W%d' % i] = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
        d['b%d' % i] = (0.1 * rng.standard_normal(n)).astype(np.float32)
    lam = np.exp(np.linspace(np.log(3500.), np.log(5900.), int(dims[-1])))
    from rvspecfit_amd.library import TemplateLibrary
    dd = dict(lam=lam, log_step=np.array(True), log_ids=np.array([0]),
              parnames=np.array(['teff', 'logg', 'feh', 'alpha'][:nd]),
              nn_dims=dims, nn_M=d['M'], nn_S=d['S'])
    for i in range(nl):
        dd['nn_W%d' % i] = d['W%d' % i]
        dd['nn_b%d' % i] = d['b%d' % i]
    lib = TemplateLibrary('nn_shapes', dd)
    for nrow in (1, 77):
        P = np.array([rng.uniform(3500, 9000, nrow), rng.uniform(0, 5, nrow),
                      rng.uniform(-2, 0, nrow), rng.uniform(0, 1, nrow)]).T[:, :nd]
        P = np.ascontiguousarray(P)
        templ, _ = lib.eval_batch(torch.as_tensor(P).to('cuda'))
        W = [(d['W%d' % i], d['b%d' % i]) for i in range(nl)]
        ref = orc.nn_forward(W, P, d['M'], d['S'])
        np.testing.assert_allclose(templ.cpu().numpy(), ref, rtol=5e-6)


# --------------------------------------------------------------------------
# vel_fit.process (Nelder-Mead in lock-step + Hessian)
# --------------------------------------------------------------------------
@pytest.fixture(scope='module')
def pcases():
    return dict(np.load(os.path.join(GOLD, 'process_cases.npz')))


def _process_args(g, t):
    pd0 = dict(zip([str(_) for _ in g[t + '/start_keys']],
                   [float(_) for _ in g[t + '/start_vals']]))
    fix = [str(_) for _ in g[t + '/fix']]
    pri = None
    if t + '/prior_keys' in g:
        pri = {str(k): tuple(v) for k, v in zip(g[t + '/prior_keys'],
                                                g[t + '/prior_vals'])}
    return pd0, fix, pri


@pytest.mark.parametrize('tag', TAGS)
def test_chisq_point_golden(cases, config, tag):
    """rvs_chisq_point (the optimiser's objective, one lane per job) against the
    reference's get_chisq values, all trials of a case in ONE call"""
    from rvspecfit_amd import spec_fit
    sds = _sds(cases, tag)
    b, _ = spec_fit.as_batch(sds)
    for npoly, rbf in ((10, True), (15, True), (5, True), (7, False)):
        sel = [i for i in range(7)
               if int(cases['%s/chisq/t%d/npoly' % (tag, i)]) == npoly and
               bool(cases['%s/chisq/t%d/rbf' % (tag, i)]) == rbf]
        for with_rot in (False, True):
            ii = [i for i in sel if np.isfinite(
                cases['%s/chisq/t%d/vsini' % (tag, i)]) == with_rot]
            if not ii:
                continue
            vel = torch.as_tensor([float(cases['%s/chisq/t%d/vel' % (tag, i)])
                                   for i in ii], dtype=torch.float64).to('cuda')
            par = torch.as_tensor(np.array(
                [cases['%s/chisq/t%d/param' % (tag, i)] for i in ii])).to('cuda')
            vs = None
            if with_rot:
                vs = torch.as_tensor([float(
                    cases['%s/chisq/t%d/vsini' % (tag, i)]) for i in ii],
                    dtype=torch.float64).to('cuda')
            idx = torch.zeros(len(ii), dtype=torch.long, device='cuda')
            with np.errstate(all='ignore'):
                c, st = spec_fit.chisq_jobs(
                    b, idx, vel, par, vs, dict(npoly=npoly, rbf_continuum=rbf),
                    config)
            for k, i in enumerate(ii):
                want = float(cases['%s/chisq/t%d/value' % (tag, i)])
                assert abs(c[k].item() / want - 1) < 1e-6, (i, c[k].item(), want)


@pytest.mark.parametrize('t', ['p0', 'p1', 'p2', 'p3'])
def test_process_golden(cases, pcases, config, t):
    """vel_fit.process against the reference's own run (scipy Nelder-Mead on the
    reference's chisq_func).  Nelder-Mead amplifies rounding differences of the
    objective into different (equally valid) paths, so the end points are
    compared at the optimiser's own tolerances (xatol 1e-2, fatol 1e-3), the
    velocity and chi^2 at the contract's (0.01 km/s, 1e-6)."""
    from rvspecfit_amd import vel_fit
    g = pcases
    sds = _sds(cases, str(g[t + '/case']))
    pd0, fix, pri = _process_args(g, t)
    cfg = dict(config)
    cfg['second_minimizer'] = False
    r = vel_fit.process(sds, pd0, fixParam=fix, options=dict(npoly=10),
                        config=cfg, priors=pri)
    assert r['minimize_success'] == bool(g[t + '/minimize_success'])
    assert abs(r['vel'] - g[t + '/vel']) < 0.01
    assert abs(r['vel_err'] / g[t + '/vel_err'] - 1) < 1e-2
    assert abs(r['chisq'] - g[t + '/chisq']) < 2e-3   # fatol-level
    assert abs(r['chisq'] / g[t + '/chisq'] - 1) < 1e-6
    names = ['teff', 'logg', 'feh', 'alpha']
    got = np.array([r['param'][_] for _ in names])
    err = g[t + '/param_err']
    ok = np.isfinite(err) & (err > 0)
    # well inside the 1-sigma uncertainty of every parameter
    assert np.all(np.abs(got - g[t + '/param'])[ok] < 0.02 * err[ok] + 1e-9)
    if np.isfinite(g[t + '/vsini']):
        assert abs(r['vsini'] - g[t + '/vsini']) < 0.05
    assert r['npix_array'] == [int(_) for _ in g[t + '/npix_array']]
    np.testing.assert_allclose(r['chisq_array'], g[t + '/chisq_array'],
                               rtol=1e-5)
    # Hessian: numdifftools' rule as restated (numdiff.py), here at the end point
    # of THIS run's simplex path; at the reference's own end point it is compared
    # to 1e-3 in test_param_uncertainties_at_reference_optimum
    assert r['bad_hessian'] == bool(g[t + '/bad_hessian'])
    gerr = np.array([r['param_err'][_] for _ in names])
    np.testing.assert_allclose(gerr[ok], err[ok], rtol=2e-2)


@pytest.mark.parametrize('t', ['p0', 'p1', 'p2', 'p3', 'p4'])
def test_param_uncertainties_at_reference_optimum(cases, pcases, config, t):
    """The Hessian stage on its own (vel_fit.py:699-725), evaluated AT the
    optimum the reference's run ended in, so that param_err / param_covar /
    bad_hessian can be compared without the optimiser's path in between:
    numdifftools' rule as restated (one exact step; the default-generator retry
    with Richardson / Wynn extrapolation on p2, whose alpha sits on the grid
    edge)."""
    from rvspecfit_amd import vel_fit
    g = pcases
    sds = _sds(cases, str(g[t + '/case']))
    pd0, fix, pri = _process_args(g, t)
    x = g[t + '/bfgs_x'] if t + '/bfgs_x' in g else g[t + '/nm_x']
    vs = float(g[t + '/vsini'])
    if not np.isfinite(vs):
        vs = pd0.get('vsini') if 'vsini' in fix else None
    r = vel_fit.param_uncertainties(sds, float(x[0]), g[t + '/param'],
                                    vsini=vs, options=dict(npoly=10),
                                    config=config, priors=pri)
    assert r['bad_hessian'] == bool(g[t + '/bad_hessian'])
    names = ['teff', 'logg', 'feh', 'alpha']
    gerr = np.array([r['param_err'][_] for _ in names])
    np.testing.assert_allclose(gerr, g[t + '/param_err'], rtol=1e-3)
    cv, cv_ref = r['param_covar'], g[t + '/param_covar']
    sc = np.sqrt(np.abs(np.outer(np.diag(cv_ref), np.diag(cv_ref))))
    assert np.max(np.abs(cv - cv_ref) / sc) < 2e-3


def test_process_nm_path_vs_oracle(cases, pcases, config, gold_libs,
                                   gold_config):
    """with the same (Cholesky) arithmetic as the C oracle the lock-step
    simplex follows scipy's path iteration for iteration on a 2-arm case"""
    from rvspecfit_amd import vel_fit
    g = pcases
    sds = _sds(cases, 'c1')
    pd0, fix, pri = _process_args(g, 'p1')
    cfg = dict(config)
    r = vel_fit.process(sds, pd0, fixParam=fix, options=dict(npoly=10),
                        config=cfg, priors=pri)
    o = orc.process(gold_specdata(cases, 'c1', orc.SpecData), pd0, fix,
                    dict(npoly=10), gold_config, gold_libs, priors=pri)
    assert abs(r['nm_nit'] - o['nm_nit'][0]) <= 0.2 * o['nm_nit'][0]
    assert abs(r['vel'] - o['vel']) < 1e-3
    assert abs(r['chisq'] - o['chisq']) < 2e-3


def test_process_batch_equals_singles(cases, pcases, config):
    from rvspecfit_amd import vel_fit
    from rvspecfit_amd.engine import SpecBatch
    lists = [_sds(cases, t) for t in ('c1', 'c3')]
    starts = [dict(teff=6000., logg=2., feh=-0.5, alpha=0.2, vsini=5.),
              dict(teff=6500., logg=2.5, feh=-0.4, alpha=0.1, vsini=20.)]
    cfg = dict(config)
    pd0 = {k: np.array([s[k] for s in starts]) for k in starts[0]}
    rb = vel_fit.process(SpecBatch.from_specdata(lists), pd0,
                         options=dict(npoly=10), config=cfg)
    for i, (sl, st) in enumerate(zip(lists, starts)):
        r1 = vel_fit.process(sl, st, options=dict(npoly=10), config=cfg)
        assert r1['nm_nit'] == int(rb['nm_nit'][i])
        assert r1['vel'] == float(rb['vel'][i])
        assert r1['chisq'] == float(rb['chisq'][i])
        assert r1['param']['teff'] == float(rb['param']['teff'][i])
        assert r1['vsini'] == float(rb['vsini'][i])
        np.testing.assert_array_equal(
            [r1['param_err'][k] for k in ('teff', 'logg', 'feh', 'alpha')],
            [rb['param_err'][k][i] for k in ('teff', 'logg', 'feh', 'alpha')])


def test_nm_bookkeeping_rows_and_pack_equal_one_block(cases, pcases, config):
    """rounds of thousands of rows run their bookkeeping as a row-parallel kernel and a
    one-block pack (option nm_split_min, default 1024 rows); forced onto every round of
    a small batch -- priors and shrinks included -- the optimiser's state is the
    one-block kernels' to the bit: iterations, evaluations, end points"""
    from rvspecfit_amd import vel_fit
    from rvspecfit_amd.engine import SpecBatch
    S = 70
    rng = np.random.RandomState(5)
    base = [_sds(cases, t) for t in ('c1', 'c3')]
    batch = SpecBatch.from_specdata([base[i % 2] for i in range(S)])
    for a in batch.arms:
        a.spec.mul_(torch.as_tensor(
            1 + 0.03 * rng.normal(size=tuple(a.spec.shape))).to(a.spec.device))
    pd0 = dict(teff=rng.uniform(5000, 6800, S), logg=rng.uniform(1.5, 4.5, S),
               feh=rng.uniform(-1.5, -0.1, S), alpha=rng.uniform(0, 0.4, S),
               vsini=rng.uniform(1, 60, S))
    out = []
    for split_min in (1 << 30, 1):
        with _lib.option('nm_split_min', split_min), vel_fit.single_stream():
            out.append(vel_fit.process(batch, dict(pd0), options=dict(npoly=10),
                                       config=dict(config)))
    a, b = out
    assert int(a['nm_nit'].max()) > 100
    for k in ('vel', 'vel_err', 'chisq', 'vsini', 'nm_nit', 'nm_nfev', 'chisq_array',
              'minimize_success'):
        assert torch.equal(a[k], b[k]), k
    for k in ('teff', 'logg', 'feh', 'alpha'):
        assert torch.equal(a['param'][k], b['param'][k]), k
        np.testing.assert_array_equal(a['param_err'][k], b['param_err'][k])
    assert a['nm_rounds'] == b['nm_rounds']


@pytest.mark.parametrize('S,spec_max', [(70, 21), (300, 64)])
def test_nm_last_rounds_in_one_launch_equal_two(cases, pcases, config, S, spec_max):
    """the optimiser's last rounds (at most nm_spec_max live simplices, and a quarter of
    the batch) evaluate all four candidate points of a step in one launch and do the
    round's bookkeeping in one kernel: iterations, evaluations (counted as scipy counts
    them), end points, status -- the two-launch rounds' to the bit"""
    from rvspecfit_amd import vel_fit
    from rvspecfit_amd.engine import SpecBatch
    rng = np.random.RandomState(17)
    base = [_sds(cases, t) for t in ('c1', 'c3')]
    batch = SpecBatch.from_specdata([base[i % 2] for i in range(S)])
    for a in batch.arms:
        a.spec.mul_(torch.as_tensor(
            1 + 0.03 * rng.normal(size=tuple(a.spec.shape))).to(a.spec.device))
    pd0 = dict(teff=rng.uniform(5000, 6800, S), logg=rng.uniform(1.5, 4.5, S),
               feh=rng.uniform(-1.5, -0.1, S), alpha=rng.uniform(0, 0.4, S),
               vsini=rng.uniform(1, 60, S))
    out = []
    for m in (0, spec_max):
        with _lib.option('nm_spec_max', m), vel_fit.single_stream():
            out.append(vel_fit.process(batch, dict(pd0), options=dict(npoly=10),
                                       config=dict(config)))
    a, b = out
    assert int(a['nm_nit'].max()) > 100
    for k in ('vel', 'vel_err', 'chisq', 'vsini', 'nm_nit', 'nm_nfev', 'chisq_array',
              'minimize_success', 'status'):
        if k in a:
            assert torch.equal(torch.as_tensor(a[k]), torch.as_tensor(b[k])), k
    for k in ('teff', 'logg', 'feh', 'alpha'):
        assert torch.equal(a['param'][k], b['param'][k]), k
        np.testing.assert_array_equal(a['param_err'][k], b['param_err'][k])

def test_process_two_halves_equal_one_batch(cases, pcases, config):
    """vel_fit.process fits a large SpecBatch as two interleaved halves on two
    streams (two host threads over the native round driver): every result is that
    of the single-stream run, bit for bit, priors and the second minimiser
    included"""
    from rvspecfit_amd import vel_fit
    from rvspecfit_amd.engine import SpecBatch
    S = 2 * vel_fit.PROCESS_SPLIT_MIN // 2 + 45
    rng = np.random.RandomState(11)
    base = [_sds(cases, t) for t in ('c1', 'c3')]
    lists = [base[i % 2] for i in range(S)]
    batch = SpecBatch.from_specdata(lists)
    # different noise per spectrum, so that the paths differ
    for a in batch.arms:
        a.spec.mul_(torch.as_tensor(
            1 + 0.02 * rng.normal(size=tuple(a.spec.shape))).to(a.spec.device))
    pd0 = dict(teff=rng.uniform(5000, 6800, S), logg=rng.uniform(1.5, 4.5, S),
               feh=rng.uniform(-1.5, -0.1, S), alpha=rng.uniform(0, 0.4, S),
               vsini=rng.uniform(1, 60, S))
    pri = {'teff': (torch.as_tensor(rng.uniform(5500, 6500, S)).to('cuda'), 300.)}
    names = ('teff', 'logg', 'feh', 'alpha')
    from rvspecfit_amd import spec_fit
    rp = {a.name: spec_fit.construct_resol_mat(a.lam_host, 2500.)
          for a in batch.arms}
    for cfg, kw in ((dict(config), {}), (dict(config, second_minimizer=True), {}),
                    (dict(config), dict(priors=pri)),
                    (dict(config), dict(resolParams=rp))):
        out = []
        for ns in (1, 2):
            vel_fit.PROCESS_STREAMS = ns
            try:
                out.append(vel_fit.process(batch, dict(pd0),
                                           options=dict(npoly=10), config=cfg,
                                           **kw))
            finally:
                vel_fit.PROCESS_STREAMS = 2
        a, b = out
        for k in ('vel', 'vel_err', 'vel_skewness', 'vel_kurtosis', 'chisq',
                  'vsini', 'nm_nit', 'nm_nfev', 'chisq_array', 'npix_array',
                  'minimize_success'):
            assert torch.equal(a[k], b[k]), k
        for k in names:
            assert torch.equal(a['param'][k], b['param'][k]), k
            np.testing.assert_array_equal(a['param_err'][k], b['param_err'][k])
        np.testing.assert_array_equal(a['bad_hessian'], b['bad_hessian'])
        for x, y in zip(a['yfit'], b['yfit']):
            assert torch.equal(x, y)
        if 'bfgs' in a:
            for k in ('nfev', 'nit', 'status'):
                np.testing.assert_array_equal(a['bfgs'][k], b['bfgs'][k])


def test_device_neldermead_equals_torch(gpu):
    """the rvs_nm_* kernels take the same path as neldermead.minimize (which the
    CPU suite pins to scipy): same nit, nfev, simplices, bit for bit -- including
    shrink steps, the maxiter exit and a 1e30 wall"""
    from refmachines import neldermead_torch as neldermead
    from rvspecfit_amd import optimizer
    rng = np.random.RandomState(3)
    for S, N, maxiter, sync in ((700, 6, 10000, 4), (64, 5, 40, 1),
                                (1500, 2, 10000, 7)):
        A = rng.normal(size=(S, N, N))
        A = np.einsum('sij,skj->sik', A, A) + np.eye(N)
        At = torch.as_tensor(A).to('cuda')
        ct = torch.as_tensor(rng.normal(size=(S, N))).to('cuda')

        def f(idx, X):
            # elementwise only: the value of a row must not depend on which
            # other rows are evaluated with it
            d = X - ct[idx]
            Ai = At[idx]
            q = torch.zeros_like(d[:, 0])
            for i in range(N):
                for k in range(N):
                    q = q + d[:, i] * Ai[:, i, k] * d[:, k]
                # a non-smooth term makes shrinks happen
                q = q + 3.0 * d[:, i].abs() + 2.0 * torch.sin(5 * d[:, i]).abs()
                q = q + 1.5 * torch.floor(3 * d[:, i]).abs()   # steps
            return torch.where(X[:, 0] > 4.0, torch.full_like(q, 1e30), q)

        simp = torch.as_tensor(rng.normal(size=(S, N + 1, N)) * 2).to('cuda')
        st0, st1 = {}, {}
        r0 = neldermead.minimize(f, simp, maxiter=maxiter, stats=st0)
        r1 = optimizer.DeviceNelderMead(S, N, 'cuda').minimize(
            optimizer.TorchObjective(f), simp, maxiter=maxiter, sync_every=sync,
            stats=st1)
        assert torch.equal(r0['nit'], r1['nit'])
        assert torch.equal(r0['nfev'], r1['nfev'])
        assert torch.equal(r0['success'], r1['success'])
        assert torch.equal(r0['final_simplex'][0], r1['final_simplex'][0])
        assert torch.equal(r0['final_simplex'][1], r1['final_simplex'][1])
        shrunk = (r0['nfev'] - (N + 1)) > 2 * (r0['nit'] - 1)
        if N > 2:
            assert bool(shrunk.any())   # the shrink branch was exercised


# --------------------------------------------------------------------------
# A9: resolution matrices (ResolMatrix, construct_resol_mat, convolve_resol)
# --------------------------------------------------------------------------
@pytest.fixture(scope='module')
def rcases():
    return dict(np.load(os.path.join(GOLD, 'resol_cases.npz')))


def _dia(g, key, n):
    import scipy.sparse
    return scipy.sparse.dia_matrix((g[key + '/data'], g[key + '/offsets']),
                                   shape=(n, n))


@pytest.mark.parametrize('tag', ['c1', 'c2'])
def test_resolution_matrix(cases, rcases, config, tag):
    from rvspecfit_amd import spec_fit
    g = rcases
    sds = _sds(cases, tag)
    opt = dict(npoly=10)
    truth = tuple(cases[tag + '/truth'])
    rp = {}
    for sd in sds:
        ref = _dia(g, '%s/rp/%s' % (tag, sd.name), len(sd.lam))
        R = spec_fit.construct_resol_mat(sd.lam, resol=2500.)
        np.testing.assert_allclose(R.mat.toarray(), ref.toarray(), rtol=1e-13,
                                   atol=1e-300)
        rp[sd.name] = R
        x = np.sin(sd.lam / 3.)
        np.testing.assert_allclose(spec_fit.convolve_resol(x, R), ref @ x,
                                   rtol=1e-12)
    # get_chisq with resol_params: grid kernel (A9 variant), all three trials
    for i in range(3):
        vs = float(g['%s/rp/t%d/vsini' % (tag, i)])
        rot = None if np.isnan(vs) else (vs, )
        val = spec_fit.get_chisq(sds, float(g['%s/rp/t%d/vel' % (tag, i)]),
                                 tuple(g['%s/rp/t%d/param' % (tag, i)]), rot,
                                 rp, options=opt, config=config)
        want = float(g['%s/rp/t%d/value' % (tag, i)])
        assert abs(val - want) < 1e-7 * max(abs(want), 1e3), (i, val, want)
    full = spec_fit.get_chisq(sds, float(cases[tag + '/vel']), truth, None, rp,
                              options=opt, config=config, full_output=True)
    assert abs(full['chisq'] / float(g[tag + '/rp/full/chisq']) - 1) < 1e-7
    np.testing.assert_allclose(full['chisq_array'],
                               g[tag + '/rp/full/chisq_array'], rtol=1e-6)
    for sd, m, rm in zip(sds, full['models'], full['raw_models']):
        np.testing.assert_allclose(rm, g['%s/rp/full/raw_model_%s'
                                         % (tag, sd.name)], rtol=1e-9)
        np.testing.assert_allclose(m, g['%s/rp/full/model_%s' % (tag, sd.name)],
                                   rtol=1e-6)
    fb = spec_fit.find_best(sds, g['vel_grid'], [truth], None, rp, options=opt,
                            config=config)
    want = g[tag + '/rp/find_best']
    assert abs(fb['best_vel'] - want[0]) < 1e-3
    assert abs(fb['vel_err'] - want[1]) < 1e-3 * max(want[1], 1)
    assert abs(fb['best_chi'] / want[2] - 1) < 1e-7
    # the point kernel (optimiser objective) with the same matrices
    b, _ = spec_fit.as_batch(sds)
    vel = torch.as_tensor([float(g['%s/rp/t%d/vel' % (tag, i)]) for i in (0, 2)],
                          dtype=torch.float64).to('cuda')
    par = torch.as_tensor(np.array([g['%s/rp/t%d/param' % (tag, i)]
                                    for i in (0, 2)])).to('cuda')
    c, st = spec_fit.chisq_jobs(b, torch.zeros(2, dtype=torch.long,
                                               device='cuda'), vel, par, None,
                                opt, config, resol_params=rp)
    for k, i in enumerate((0, 2)):
        want = float(g['%s/rp/t%d/value' % (tag, i)])
        assert abs(c[k].item() - want) < 1e-7 * max(abs(want), 1e3)
    # per-spectrum matrices (SpecData.resolution) + continuum
    sds2 = [spec_fit.SpecData(sd.name, sd.lam, sd.spec, sd.espec,
                              badmask=sd.badmask,
                              resolution=spec_fit.ResolMatrix(
                                  _dia(g, '%s/own/%s' % (tag, sd.name),
                                       len(sd.lam)))) for sd in sds]
    for i in range(3):
        vs = float(g['%s/rp/t%d/vsini' % (tag, i)])
        rot = None if np.isnan(vs) else (vs, )
        val = spec_fit.get_chisq(sds2, float(g['%s/rp/t%d/vel' % (tag, i)]),
                                 tuple(g['%s/rp/t%d/param' % (tag, i)]), rot,
                                 options=opt, config=config)
        want = float(g['%s/own/t%d/value' % (tag, i)])
        assert abs(val - want) < 1e-7 * max(abs(want), 1e3), (i, val, want)
    cc = spec_fit.get_chisq_continuum(sds2, options=opt)
    np.testing.assert_allclose(cc['chisq_array'],
                               g[tag + '/own/cont/chisq_array'], rtol=1e-8)
    fb = spec_fit.find_best(sds2, g['vel_grid'], [truth], (30., ), options=opt,
                            config=config)
    want = g[tag + '/own/find_best']
    assert abs(fb['best_vel'] - want[0]) < 1e-3
    assert abs(fb['best_chi'] / want[2] - 1) < 1e-7
    with pytest.raises(ValueError):
        spec_fit.get_chisq(sds2, 0., truth, None, rp, options=opt, config=config)


@pytest.mark.parametrize('tag', ['c1', 'c2'])
def test_fast_interp_and_espec_dict(cases, rcases, config, tag):
    """get_chisq(fast_interp=True) (nearest template pixel, spec_fit.py:913-918)
    and a per-setup espec_systematic dict (spec_fit.py:933-937)"""
    from rvspecfit_amd import spec_fit
    g = rcases
    sds = _sds(cases, tag)
    opt = dict(npoly=10)
    for i in range(3):
        vs = float(g['%s/rp/t%d/vsini' % (tag, i)])
        rot = None if np.isnan(vs) else (vs, )
        val = spec_fit.get_chisq(sds, float(g['%s/rp/t%d/vel' % (tag, i)]),
                                 tuple(g['%s/rp/t%d/param' % (tag, i)]), rot,
                                 options=opt, config=config, fast_interp=True)
        want = float(g['%s/fast/t%d/value' % (tag, i)])
        assert abs(val - want) < 1e-7 * max(abs(want), 1e3), (i, val, want)
    full = spec_fit.get_chisq(sds, float(cases[tag + '/vel']),
                              tuple(cases[tag + '/truth']), None, options=opt,
                              config=config, fast_interp=True, full_output=True)
    np.testing.assert_allclose(full['chisq_array'],
                               g[tag + '/fast/full/chisq_array'], rtol=1e-6)
    esd = {sd.name: float(v) for sd, v in zip(sds, g[tag + '/esys_dict/vals'])}
    val = spec_fit.get_chisq(sds, float(cases[tag + '/vel']),
                             tuple(cases[tag + '/truth']), None, options=opt,
                             config=config, espec_systematic=esd)
    want = float(g[tag + '/esys_dict/value'])
    assert abs(val - want) < 1e-7 * max(abs(want), 1e3)


# --------------------------------------------------------------------------
# Delaunay (triangulation) evaluator: spec_inter.TriInterp -> rvs_template_tri
# --------------------------------------------------------------------------
def test_triangulation(cases, gpu):
    from rvspecfit_amd import spec_fit, spec_inter
    from rvspecfit_amd.library import TemplateLibrary
    g = dict(np.load(os.path.join(GOLD, 'tri_cases.npz')))
    cfg = dict(GOLD_CONFIG)
    cfg['template_lib'] = 'golden-tri://'
    olibs = {}
    for n in ('gold_b', 'gold_r'):
        d = np.load(os.path.join(GOLD, 'lib_tri_%s.npz' % n))
        spec_inter.register_library(TemplateLibrary(n, d), 'golden-tri://')
        olibs[n] = orc.TriLibrary(d)
    P = g['params']
    for n in ('gold_b', 'gold_r'):
        it = spec_inter.getInterpolator(n, cfg)
        assert it.lib.kind == 'triangulation'
        with np.errstate(all='ignore'):
            templ, outside, sx, wts = it.lib.eval_batch(
                torch.as_tensor(P).to('cuda'), details=True)
        templ, outside = templ.cpu().numpy(), outside.cpu().numpy()
        sx = sx.cpu().numpy()
        for i, p in enumerate(P):
            ref_sx = int(g[n + '/simplex'][i])
            if ref_sx < 0:
                assert sx[i] == 0x7fffffff
                assert np.isnan(outside[i]) and np.isnan(templ[i]).all()
                continue
            # integer work: the simplex of the exhaustive search (oracle); the
            # reference's walk may stop in a neighbour on a shared face, where
            # the interpolant is the same
            with np.errstate(all='ignore'):
                _, info = olibs[n].eval(p, details=True)
            assert sx[i] == info['simplex']
            np.testing.assert_allclose(templ[i], g[n + '/eval'][i], rtol=1e-12)
            assert abs(outside[i] - g[n + '/outside'][i]) < 1e-12
    sds = _sds(cases, 'c1')
    for i in range(4):
        vs = float(g['c1/t%d/vsini' % i])
        with np.errstate(all='ignore'):
            val = spec_fit.get_chisq(sds, float(g['c1/t%d/vel' % i]),
                                     tuple(g['c1/t%d/param' % i]),
                                     None if np.isnan(vs) else (vs, ),
                                     options=dict(npoly=10), config=cfg)
        want = float(g['c1/t%d/value' % i])
        assert abs(val - want) < 1e-7 * max(abs(want), 1e3), (i, val, want)
    truth = tuple(cases['c1/truth'])
    fb = spec_fit.find_best(sds, g['vel_grid'], [truth, tuple(P[3]), tuple(P[0])],
                            None, options=dict(npoly=10), config=cfg)
    want = g['c1/find_best']
    assert abs(fb['best_vel'] - want[0]) < 1e-3
    assert abs(fb['best_chi'] / want[2] - 1) < 1e-7
    np.testing.assert_allclose(fb['best_param'], g['c1/best_param'])


def test_process_on_delaunay_libraries_rounds_in_c(cases, config, monkeypatch):
    """vel_fit.process of a batch on Delaunay libraries: the optimiser's rounds inside
    the library (rvs_nm_run / rvs_bfgs_run with rvs_nm_objective.tri: find_simplex
    through the bucket grid + blend per arm, then rvs_objective_from_template) against
    the same rounds driven from Python on the same kernels (NATIVE_ROUNDS off, the
    host BFGS machines) -- every number bit for bit through the simplex stage, the
    BFGS polish to its own tolerance"""
    from rvspecfit_amd import optimizer, spec_inter, vel_fit
    from rvspecfit_amd.engine import SpecBatch
    from rvspecfit_amd.library import TemplateLibrary
    cfg = dict(config, template_lib='golden-tri://')
    for n in ('gold_b', 'gold_r'):
        d = np.load(os.path.join(GOLD, 'lib_tri_%s.npz' % n))
        spec_inter.register_library(TemplateLibrary(n, d), 'golden-tri://')
    rng = np.random.RandomState(12)
    S = 30
    lists = [_sds(cases, ('c1', 'c3')[i % 2]) for i in range(S)]
    batch = SpecBatch.from_specdata(lists)
    for a in batch.arms:
        a.spec.mul_(torch.as_tensor(
            1 + 0.02 * rng.normal(size=tuple(a.spec.shape))).to(a.spec.device))
    pd0 = dict(teff=rng.uniform(5200, 6500, S), logg=rng.uniform(2., 4., S),
               feh=rng.uniform(-1.2, -0.3, S), alpha=rng.uniform(0.1, 0.3, S),
               vsini=rng.uniform(1, 60, S))
    out = {}
    for sm in (False, True):
        c2 = dict(cfg, second_minimizer=sm)
        for native in (True, False):
            monkeypatch.setattr(optimizer, 'NATIVE_ROUNDS', native)
            with np.errstate(all='ignore'):
                out[sm, native] = vel_fit.process(batch, dict(pd0),
                                                  options=dict(npoly=10), config=c2)
    a, b = out[False, True], out[False, False]
    assert int(a['nm_nit'].max()) > 50
    for k in ('vel', 'chisq', 'vsini', 'nm_nit', 'nm_nfev', 'nm_vel'):
        assert torch.equal(a[k], b[k]), k
    for k in ('teff', 'logg', 'feh', 'alpha'):
        assert torch.equal(a['param'][k], b['param'][k]), k
    a, b = out[True, True], out[True, False]
    assert a['bfgs']['device'] and not b['bfgs']['device']
    assert torch.equal(a['nm_nit'], b['nm_nit'])
    assert (a['chisq'] - b['chisq']).abs().max().item() < 2e-3


@pytest.mark.parametrize('kind,snr,feh0', [('regulargrid', 100., -0.2),
                                           ('triangulation', 1000., 0.)])
def test_fit_fake(cases, config, kind, snr, feh0):
    """the reference's own end-to-end pins, tests/test_fit_fake_grid.py:51
    (regular grid, S/N 100) and tests/test_fit_fake.py:48 (Delaunay, S/N 1000):
    a synthetic star at (5000, 2, -1, 0.2), v0 ~ N(0, 100), vel_fit.process from
    (teff 5000, logg 2, feh feh0, alpha 0.2, vsini 0.1), npoly 15;
    |vel - v0| < max(10, 3 vel_err)."""
    from rvspecfit_amd import spec_fit, spec_inter, synth, vel_fit
    from rvspecfit_amd.library import TemplateLibrary
    cfg = dict(config)
    if kind == 'triangulation':
        cfg['template_lib'] = 'golden-tri://'
        d = np.load(os.path.join(GOLD, 'lib_tri_gold_b.npz'))
        spec_inter.register_library(TemplateLibrary('gold_b', d), 'golden-tri://')
    lam = cases['c0/gold_b/lam']
    for seed in (1, 2, 3):
        rng = np.random.RandomState(seed)
        v0 = rng.normal(0, 100)
        spec, espec = synth.fake_observation(lam, 5000., 2., -1., 0.2, v0, snr,
                                             rng, wresol=4700. / 2000 / 2.35)
        sd = [spec_fit.SpecData('gold_b', lam, spec, espec)]
        res = vel_fit.process(sd, dict(logg=2, teff=5000, feh=feh0, alpha=0.2,
                                       vsini=0.1), fixParam=[],
                              config=cfg, options=dict(npoly=15))
        assert abs(res['vel'] - v0) < max(10, 3 * res['vel_err']), \
            (seed, res['vel'], v0, res['vel_err'])
        assert res['minimize_success']
        assert len(res['yfit'][0]) == len(lam)


def test_nm_round_drivers_agree(cases, config):
    """the Nelder-Mead rounds of a fused objective driven from C (rvs_nm_run,
    the default) and from Python (optimizer.NATIVE_ROUNDS = False, the loop every
    non-fused objective takes): same launches in the same order, so the same
    simplices bit for bit"""
    from rvspecfit_amd import optimizer, vel_fit
    from rvspecfit_amd.engine import SpecBatch
    rng = np.random.RandomState(5)
    S = 40
    lists = [_sds(cases, ('c1', 'c3')[i % 2]) for i in range(S)]
    batch = SpecBatch.from_specdata(lists)
    for a in batch.arms:
        a.spec.mul_(torch.as_tensor(
            1 + 0.02 * rng.normal(size=tuple(a.spec.shape))).to(a.spec.device))
    pd0 = dict(teff=rng.uniform(5000, 6800, S), logg=rng.uniform(1.5, 4.5, S),
               feh=rng.uniform(-1.5, -0.1, S), alpha=rng.uniform(0, 0.4, S),
               vsini=rng.uniform(1, 60, S))
    out = []
    for native in (True, False):
        optimizer.NATIVE_ROUNDS = native
        try:
            out.append(vel_fit.process(batch, dict(pd0), options=dict(npoly=10),
                                       config=dict(config)))
        finally:
            optimizer.NATIVE_ROUNDS = True
    a, b = out
    for k in ('vel', 'chisq', 'vsini', 'nm_nit', 'nm_nfev', 'nm_vel'):
        assert torch.equal(a[k], b[k]), k
    for k in ('teff', 'logg', 'feh', 'alpha'):
        assert torch.equal(a['param'][k], b['param'][k]), k
        np.testing.assert_array_equal(a['param_err'][k], b['param_err'][k])


def test_process_bfgs_implementations_agree(cases, config, monkeypatch):
    """second_minimizer through the C++ coroutines (what the product runs) and
    through the Python generators that the CPU suite pins to scipy
    (tests/refmachines/bfgs_scipy_restated.py, put in the driver's place): the
    real objective is noisy at the gradient step, so the two follow each other to
    rounding, not to the bit -- same exit statistics, chi^2 and parameters well
    inside the optimiser's tolerances"""
    from refmachines import bfgs_scipy_restated as bfgs_ref
    from rvspecfit_amd import bfgs, vel_fit
    from rvspecfit_amd.engine import SpecBatch
    rng = np.random.RandomState(6)
    S = 24
    lists = [_sds(cases, ('c1', 'c3')[i % 2]) for i in range(S)]
    batch = SpecBatch.from_specdata(lists)
    for a in batch.arms:
        a.spec.mul_(torch.as_tensor(
            1 + 0.02 * rng.normal(size=tuple(a.spec.shape))).to(a.spec.device))
    pd0 = dict(teff=rng.uniform(5000, 6800, S), logg=rng.uniform(1.5, 4.5, S),
               feh=rng.uniform(-1.5, -0.1, S), alpha=rng.uniform(0, 0.4, S),
               vsini=rng.uniform(1, 60, S))
    cfg = dict(config, second_minimizer=True)
    out = {}
    out['native'] = vel_fit.process(batch, dict(pd0), options=dict(npoly=10),
                                    config=cfg)
    monkeypatch.setattr(bfgs, 'minimize_lockstep_native',
                        bfgs_ref.minimize_lockstep)
    out['python'] = vel_fit.process(batch, dict(pd0), options=dict(npoly=10),
                                    config=cfg)
    a, b = out['native'], out['python']
    assert a['second_minimizer_run'] and b['second_minimizer_run']
    assert torch.equal(a['nm_nit'], b['nm_nit'])      # same simplex stage
    assert (a['chisq'] - b['chisq']).abs().max().item() < 2e-3    # fatol level
    # the velocity at equal chi^2 (to 2e-3) along a flat direction: a few per cent
    # of its own uncertainty (vel_err is ~1 km/s for these spectra), 0.01 km/s for
    # most -- which spectra sit at a few 1e-2 depends on the last bits of the
    # objective (it moved when the continuum basis became rvs_basis_build's)
    dv = (a['vel'] - b['vel']).abs()
    assert (dv <= torch.clamp(0.05 * torch.as_tensor(a['vel_err']).to(dv.device),
                              min=1e-2)).all(), dv
    assert (dv < 1e-2).float().mean().item() > 0.8
    assert abs(np.mean(a['bfgs']['nfev']) / np.mean(b['bfgs']['nfev']) - 1) < 0.5


@pytest.mark.parametrize('S', [24, 700])
def test_process_bfgs_device_equals_host(cases, config, monkeypatch, S):
    """second_minimizer with its rounds on the device (rvs_bfgs_run: one thread per
    run of csrc/bfgs_machine.h, the requests gathered by kernels, the objective of
    rvs_nm_run in chunks of S rows) against the same machines on the host around the
    Python objective (rvs_bfgs_begin / _pending / _feed, the pair the CPU suite pins
    to scipy): one source for both, the same objective kernel behind both -- the
    runs take the same path (nit, nfev, status) and end in the same point.  700
    spectra: a half of the batch is 350 runs = 6 waves of the advance kernel, a
    first round of 7 chunks."""
    from rvspecfit_amd import vel_fit
    from rvspecfit_amd.engine import SpecBatch
    rng = np.random.RandomState(16)
    lists = [_sds(cases, ('c1', 'c3')[i % 2]) for i in range(S)]
    batch = SpecBatch.from_specdata(lists)
    for a in batch.arms:
        a.spec.mul_(torch.as_tensor(
            1 + 0.02 * rng.normal(size=tuple(a.spec.shape))).to(a.spec.device))
    pd0 = dict(teff=rng.uniform(5000, 6800, S), logg=rng.uniform(1.5, 4.5, S),
               feh=rng.uniform(-1.5, -0.1, S), alpha=rng.uniform(0, 0.4, S),
               vsini=rng.uniform(1, 60, S))
    cfg = dict(config, second_minimizer=True)
    out = {}
    for name, flag in (('device', True), ('host', False)):
        monkeypatch.setattr(vel_fit, 'BFGS_ON_DEVICE', flag)
        out[name] = vel_fit.process(batch, dict(pd0), options=dict(npoly=10),
                                    config=cfg)
    a, b = out['device'], out['host']
    assert a['bfgs']['device'] and not b['bfgs']['device']
    assert torch.equal(a['nm_nit'], b['nm_nit'])      # same simplex stage
    same = (a['bfgs']['nit'] == b['bfgs']['nit']) & \
        (a['bfgs']['nfev'] == b['bfgs']['nfev']) & \
        (a['bfgs']['status'] == b['bfgs']['status'])
    # (the machines differ in pow() of _cubicmin and in the order in which the
    # prior / vsini penalties join the sum: a run whose zoom step or tie falls on
    # that last bit may part ways)
    assert same.mean() > 0.9, (same.mean(), a['bfgs']['nfev'][~same],
                               b['bfgs']['nfev'][~same])
    assert a['bfgs']['nfev'].max() > 20
    sm = torch.as_tensor(same).to(a['vel'].device)
    for k in ('vel', 'chisq'):
        assert torch.equal(a[k][sm], b[k][sm]), k
    for k in ('teff', 'logg', 'feh', 'alpha'):
        assert torch.equal(a['param'][k][sm], b['param'][k][sm]), k
    assert (a['chisq'] - b['chisq']).abs().max().item() < 2e-3    # fatol level


@pytest.mark.parametrize('t', ['p2', 'p3'])
def test_process_bfgs_device_with_priors_and_fixed_parameters(cases, pcases, config,
                                                              monkeypatch, t):
    """the polish on the device with what the parameter mapping can carry: p2 -- alpha
    fixed, a Normal prior on feh (vel_fit.py:205-226: the prior joins the objective
    the minimisers see); p3 -- vsini given and fixed.  A batch of eight copies with
    different noise: the device loop (rvs_bfgs_run on rvs_proc_map's mapping) and the
    host machines around the Python objective take the same path."""
    from rvspecfit_amd import vel_fit
    from rvspecfit_amd.engine import SpecBatch
    g = pcases
    sds = _sds(cases, str(g[t + '/case']))
    pd0, fix, pri = _process_args(g, t)
    S = 8
    rng = np.random.RandomState(31)
    batch = SpecBatch.from_specdata([sds] * S)
    for a in batch.arms:
        a.spec.mul_(torch.as_tensor(
            1 + 0.01 * rng.normal(size=tuple(a.spec.shape))).to(a.spec.device))
    pdb = {k: np.full(S, v) for k, v in pd0.items()}
    prb = None if pri is None else {k: (torch.full((S, ), float(m), dtype=torch.float64,
                                                   device='cuda'),
                                        torch.full((S, ), float(sg),
                                                   dtype=torch.float64, device='cuda'))
                                    for k, (m, sg) in pri.items()}
    cfg = dict(config, second_minimizer=True)
    out = {}
    for name, flag in (('device', True), ('host', False)):
        monkeypatch.setattr(vel_fit, 'BFGS_ON_DEVICE', flag)
        out[name] = vel_fit.process(batch, dict(pdb), fixParam=fix,
                                    options=dict(npoly=10), config=cfg, priors=prb)
    a, b = out['device'], out['host']
    assert a['bfgs']['device'] and not b['bfgs']['device']
    assert torch.equal(a['nm_nit'], b['nm_nit'])
    same = (a['bfgs']['nit'] == b['bfgs']['nit']) & \
        (a['bfgs']['nfev'] == b['bfgs']['nfev']) & \
        (a['bfgs']['status'] == b['bfgs']['status'])
    assert same.sum() >= S - 1, (a['bfgs']['nfev'], b['bfgs']['nfev'])
    assert (a['chisq'] - b['chisq']).abs().max().item() < 2e-3
    assert (a['vel'] - b['vel']).abs().max().item() < 1e-2
    # the fixed parameter stays where it was put
    if 'alpha' in fix:
        assert torch.equal(a['param']['alpha'], torch.full_like(a['param']['alpha'],
                                                                pd0['alpha']))
    if 'vsini' in fix:
        assert 'vsini' not in a or a['vsini'] is None or \
            torch.equal(a['vsini'], torch.full_like(a['vsini'], pd0['vsini']))


@pytest.mark.parametrize('second', [False, True])
def test_process_early_split_equals_unsplit(cases, config, monkeypatch, second):
    """vel_fit.process lets the spectra that leave the simplex stage first go on to
    their BFGS polish, refinement and Hessian on a second stream while the stragglers'
    last rounds run (rvs_nm_run returns at stop_below running simplices and is called
    again for the rest): every number equals the run in which all spectra wait for the
    slowest simplex, bit for bit -- no spectrum sees another, and a resumed run
    continues where it stopped."""
    from rvspecfit_amd import vel_fit
    from rvspecfit_amd.engine import SpecBatch
    rng = np.random.RandomState(21)
    S = 400
    lists = [_sds(cases, ('c1', 'c3')[i % 2]) for i in range(S)]
    batch = SpecBatch.from_specdata(lists)
    for a in batch.arms:
        a.spec.mul_(torch.as_tensor(
            1 + 0.02 * rng.normal(size=tuple(a.spec.shape))).to(a.spec.device))
    pd0 = dict(teff=rng.uniform(5000, 6800, S), logg=rng.uniform(1.5, 4.5, S),
               feh=rng.uniform(-1.5, -0.1, S), alpha=rng.uniform(0, 0.4, S),
               vsini=rng.uniform(1, 60, S))
    cfg = dict(config, second_minimizer=second)
    out = {}
    for flag in (True, False):
        monkeypatch.setattr(vel_fit, 'EARLY_SPLIT', flag)
        del vel_fit.EARLY_SPLITS[:]
        out[flag] = vel_fit.process(batch, dict(pd0), options=dict(npoly=10),
                                    config=cfg)
        # (two halves of 200 spectra: each split once)
        assert (len(vel_fit.EARLY_SPLITS) == 2) == flag, vel_fit.EARLY_SPLITS
        if flag:
            assert all(f >= 32 and r >= 1 for f, r in vel_fit.EARLY_SPLITS)
    a, b = out[True], out[False]
    for k in ('vel', 'chisq', 'vsini', 'nm_nit', 'nm_nfev', 'nm_vel', 'vel_err',
              'minimize_success', 'status'):
        assert torch.equal(torch.as_tensor(a[k]), torch.as_tensor(b[k])), k
    for k in ('teff', 'logg', 'feh', 'alpha'):
        assert torch.equal(a['param'][k], b['param'][k]), k
        np.testing.assert_array_equal(a['param_err'][k], b['param_err'][k])
    np.testing.assert_array_equal(a['bad_hessian'], b['bad_hessian'])
    for m1, m2 in zip(a['yfit'], b['yfit']):
        assert torch.equal(m1, m2)
    assert a['objective_evals'] == b['objective_evals']
    if second:
        for k in ('nit', 'nfev', 'status'):
            np.testing.assert_array_equal(a['bfgs'][k], b['bfgs'][k])


@pytest.mark.parametrize('S', [40, 1300])
def test_nm_round_kernels_equal_chain(cases, config, S):
    """rvs_nm_run's rounds -- three bookkeeping kernels that also sum the arms and
    map the next rows, the objective skipping the rows behind the device counts --
    against the same rounds as a chain of the stand-alone kernels (option nm_glue = 0):
    every number of vel_fit.process bit for bit.  1300 spectra: a half of the batch is
    650 rows, which the one-block kernels that hold a simplex in registers (512
    threads) take in two trips -- the second trip reads job tables of the evaluation
    just done while the first trip's rows are already mapped for the next one."""
    from rvspecfit_amd import vel_fit
    from rvspecfit_amd.engine import SpecBatch
    rng = np.random.RandomState(8)
    lists = [_sds(cases, ('c1', 'c3')[i % 2]) for i in range(S)]
    batch = SpecBatch.from_specdata(lists)
    for a in batch.arms:
        a.spec.mul_(torch.as_tensor(
            1 + 0.03 * rng.normal(size=tuple(a.spec.shape))).to(a.spec.device))
    pd0 = dict(teff=rng.uniform(5000, 6800, S), logg=rng.uniform(1.5, 4.5, S),
               feh=rng.uniform(-1.5, -0.1, S), alpha=rng.uniform(0, 0.4, S),
               vsini=rng.uniform(1, 60, S))
    out = {}
    for glue in ('1', '0'):
        with _lib.option('nm_glue', int(glue)):
            out[glue] = vel_fit.process(batch, dict(pd0), options=dict(npoly=10),
                                        config=config)
    a, b = out['1'], out['0']
    assert torch.equal(a['nm_nit'], b['nm_nit'])
    assert torch.equal(a['nm_nfev'], b['nm_nfev'])
    assert int(a['nm_nit'].max()) > 50
    for k in ('vel', 'chisq', 'vel_err'):
        assert np.array_equal(np.asarray(torch.as_tensor(a[k]).cpu()),
                              np.asarray(torch.as_tensor(b[k]).cpu())), k
    for k in a['param']:
        assert np.array_equal(np.asarray(torch.as_tensor(a['param'][k]).cpu()),
                              np.asarray(torch.as_tensor(b['param'][k]).cpu())), k


def test_process_second_minimizer(cases, pcases, config):
    """config second_minimizer=True (the reference default): BFGS from the
    simplex optimum.  The reference's own BFGS (golden p4; scipy 1.7 there,
    1.15 here) ends in 'precision loss' after moving chi^2 by 1e-11; ours must
    do the same kind of nothing: not increase chi^2, stay within the contract."""
    from rvspecfit_amd import vel_fit
    g = pcases
    sds = _sds(cases, 'c1')
    pd0, fix, pri = _process_args(g, 'p4')
    cfg = dict(config)
    cfg['second_minimizer'] = False
    r0 = vel_fit.process(sds, pd0, fixParam=fix, options=dict(npoly=10),
                         config=cfg, priors=pri)
    cfg['second_minimizer'] = True
    r1 = vel_fit.process(sds, pd0, fixParam=fix, options=dict(npoly=10),
                         config=cfg, priors=pri)
    assert r1['second_minimizer_run'] and not r0['second_minimizer_run']
    assert r1['bfgs']['status'][0] in (0, 2)
    assert r1['chisq'] <= r0['chisq'] + 1e-6
    assert abs(r1['vel'] - g['p4/vel']) < 0.01
    assert abs(r1['chisq'] / g['p4/chisq'] - 1) < 1e-6
    err = g['p4/param_err']
    got = np.array([r1['param'][_] for _ in ('teff', 'logg', 'feh', 'alpha')])
    assert np.all(np.abs(got - g['p4/param']) < 0.02 * err)


@pytest.mark.parametrize('tag', TAGS)
def test_objective_fused(cases, config, tag):
    """rvs_objective_fused (gather + FIR + spline solve + chi^2 in one kernel,
    template in LDS) against the chain of stand-alone kernels and the
    reference's get_chisq values"""
    from rvspecfit_amd import engine, spec_fit
    sds = _sds(cases, tag)
    b, _ = spec_fit.as_batch(sds)
    for npoly, rbf in ((10, True), (15, True), (5, True), (7, False)):
        for with_rot in (False, True):
            ii = [i for i in range(7)
                  if int(cases['%s/chisq/t%d/npoly' % (tag, i)]) == npoly and
                  bool(cases['%s/chisq/t%d/rbf' % (tag, i)]) == rbf and
                  np.isfinite(cases['%s/chisq/t%d/vsini' % (tag, i)]) == with_rot]
            if not ii:
                continue
            vel = torch.as_tensor([float(cases['%s/chisq/t%d/vel' % (tag, i)])
                                   for i in ii], dtype=torch.float64).to('cuda')
            par = torch.as_tensor(np.array(
                [cases['%s/chisq/t%d/param' % (tag, i)] for i in ii])).to('cuda')
            vs = None
            if with_rot:
                vs = torch.as_tensor([float(
                    cases['%s/chisq/t%d/vsini' % (tag, i)]) for i in ii],
                    dtype=torch.float64).to('cuda')
            idx = torch.zeros(len(ii), dtype=torch.long, device='cuda')
            opt = dict(npoly=npoly, rbf_continuum=rbf)
            out = {}
            for fused in (True, False):
                engine.FUSED_OBJECTIVE = fused
                try:
                    with np.errstate(all='ignore'):
                        out[fused] = spec_fit.chisq_jobs(b, idx, vel, par, vs,
                                                         opt, config)
                finally:
                    engine.FUSED_OBJECTIVE = True
            c1, s1 = out[True]
            c0, s0 = out[False]
            assert torch.equal(s0, s1)
            for k, i in enumerate(ii):
                want = float(cases['%s/chisq/t%d/value' % (tag, i)])
                sc = max(abs(want), 1e3)
                assert abs(c1[k].item() - c0[k].item()) < 1e-11 * sc, (i, k)
                assert abs(c1[k].item() - want) < 1e-6 * sc, (i, c1[k].item())


@pytest.mark.parametrize('ndim', [3, 2])
def test_objective_fused_on_lower_dimensional_grids(cases, ndim):
    """grids of fewer than four dimensions under the one-kernel objective: the gather
    keeps sixteen vertex slots in registers, the slots behind the grid's 2^ndim carry
    weight zero and read the last vertex row again -- against the chain of stand-alone
    kernels, in-grid points, points on the grid's edge and outside it"""
    from rvspecfit_amd import engine, spec_fit, spec_inter
    from rvspecfit_amd.library import TemplateLibrary
    root = 'golden-ndim%d://' % ndim
    for n in ('gold_b', 'gold_r'):
        d = gold_lib_dict(n)
        idg = np.asarray(d['idgrid'])
        sl = (Ellipsis, ) + (1, ) * (4 - ndim)        # fix the last dimensions
        sub = idg[sl]
        rows = sub.ravel()
        keep = rows[rows >= 0]
        new = np.full(rows.shape, -1, dtype=np.int64)
        new[rows >= 0] = np.arange(len(keep))
        dd = {k: v for k, v in d.items() if not k.startswith('uvec')}
        dd.update(dats=np.asarray(d['dats'])[keep], vec=np.asarray(d['vec'])[:ndim, keep],
                  idgrid=new.reshape(sub.shape), parnames=np.asarray(d['parnames'])[:ndim])
        for i in range(ndim):
            dd['uvec%d' % i] = d['uvec%d' % i]
        spec_inter.register_library(TemplateLibrary(n, dd), root)
    cfg = dict(GOLD_CONFIG)
    cfg['template_lib'] = root
    b, _ = spec_fit.as_batch(_sds(cases, 'c1'))
    rng = np.random.RandomState(ndim)
    J = 40
    lo, hi = [3500., 0.5, -2.0][:ndim], [9000., 4.5, 0.0][:ndim]
    par = np.stack([rng.uniform(l, h, J) for l, h in zip(lo, hi)], 1)
    par[::9, 0] = 20000.0                             # outside: nearest neighbour
    par = torch.as_tensor(par).to('cuda')
    vel = torch.as_tensor(rng.uniform(-200, 200, J)).to('cuda')
    vs = torch.as_tensor(rng.uniform(0, 80, J)).to('cuda')
    idx = torch.zeros(J, dtype=torch.long, device='cuda')
    out = {}
    for fused in (True, False):
        engine.FUSED_OBJECTIVE = fused
        try:
            with np.errstate(all='ignore'):
                out[fused] = spec_fit.chisq_jobs(b, idx, vel, par, vs,
                                                 dict(npoly=10), cfg)
        finally:
            engine.FUSED_OBJECTIVE = True
    (c1, s1), (c0, s0) = out[True], out[False]
    assert torch.equal(s0, s1)
    assert torch.isfinite(c1).all()
    sc = torch.clamp(c0.abs(), min=1e3)
    assert float(((c1 - c0).abs() / sc).max()) < 1e-11


@pytest.mark.parametrize('ntp', [33, 262, 6215, 6658, 8192])
def test_spline_factors_chunk_order(gpu, ntp):
    """rvs_spline_factors lays the factors out a second time in the order the fused
    objective kernel's 512 threads own their rows (csrc/common.h: CH rows per thread,
    odd where it can be): records {1/h_u, 1/h_{u+1}, g_u, e_u} at [q][t] for row u =
    t CH + q, then the backward multipliers in pairs of rows -- bit for bit the five
    arrays in front of them, zero where there is no such row"""
    lam = np.exp(np.linspace(np.log(3600.), np.log(5800.), ntp))
    kn = torch.as_tensor(lam).to('cuda')
    n = _lib.lib().rvs_spline_factors_len(ntp)
    assert n == 5 * ntp + 5 * 8192
    fac = torch.full((n, ), np.nan, dtype=torch.float64, device='cuda')
    assert _lib.lib().rvs_spline_factors(_lib.ptr(kn), ntp, _lib.ptr(fac),
                                         _lib.stream()) == 0
    f = fac.cpu().numpy()
    g, e, cc, hh, ih = (f[i * ntp:(i + 1) * ntp] for i in range(5))
    np.testing.assert_array_equal(hh[:-1], np.diff(lam))
    m = ntp - 2
    ch = max(12, -(-m // 512))
    if (ch | 1) <= 16:
        ch |= 1
    assert ch % 2 == 1 or ch == 16
    rec = f[5 * ntp:5 * ntp + 4 * 8192].reshape(16, 512, 4)
    ccp = f[5 * ntp + 4 * 8192:].reshape(8, 512, 2)
    u = np.arange(512)[None, :] * ch + np.arange(16)[:, None]      # [q, t]
    ok = (np.arange(16)[:, None] < ch) & (u < m)
    uc = np.where(ok, u, 0)
    for a, src in enumerate((ih[uc], ih[uc + 1], g[uc], e[uc])):
        np.testing.assert_array_equal(rec[:, :, a], np.where(ok, src, 0.0))
    want_c = np.where(ok, cc[uc], 0.0)                              # [q, t]
    np.testing.assert_array_equal(ccp[:, :, 0], want_c[0::2])
    np.testing.assert_array_equal(ccp[:, :, 1], want_c[1::2])


@pytest.mark.parametrize('ntp', [31, 33, 64, 262, 263, 700])
def test_objective_fused_short_templates(ntp):
    """template grids of a few dozen to a few hundred knots under the one-kernel
    objective: the FIR window and the spline chunks read a fixed number of doubles
    behind a thread's own rows (the launcher asks for 32 knots; 31 goes the kernel
    chain's way), chunks of 13 rows leave most of the 512 threads without rows, 262 /
    263 knots end on a full chunk / start one more -- against the chain of stand-alone
    kernels"""
    from rvspecfit_amd import engine, spec_fit, spec_inter
    from rvspecfit_amd.library import TemplateLibrary
    root = 'golden-short%d://' % ntp
    d = gold_lib_dict('gold_b')
    lam = np.asarray(d['lam'])[100:100 + ntp]
    dd = dict(d)
    dd.update(lam=lam, dats=np.ascontiguousarray(np.asarray(d['dats'])[:, 100:100 + ntp]))
    dd = {k: v for k, v in dd.items() if not k.startswith('ccf')}
    lib = TemplateLibrary('gold_b', dd)
    spec_inter.register_library(lib, root)
    cfg = dict(GOLD_CONFIG, template_lib=root)
    rng = np.random.RandomState(ntp)
    npix = max(8, ntp // 3)
    wl = np.linspace(lam[6] * 1.001, lam[-7] * 0.999, npix)
    sd = [spec_fit.SpecData('gold_b', wl, 1 + 0.1 * rng.standard_normal(npix),
                            np.full(npix, 0.1))]
    b, _ = spec_fit.as_batch(sd)
    libs = spec_inter.get_libs(b.names, cfg)
    assert engine.can_fuse_objective(b, libs, None, npoly=5) == (ntp >= 32)
    J = 24
    par = torch.as_tensor(np.stack([rng.uniform(4600, 7000, J), rng.uniform(1., 4., J),
                                    rng.uniform(-1.8, -0.1, J),
                                    rng.uniform(0.0, 0.4, J)], 1)).to('cuda')
    vel = torch.as_tensor(rng.uniform(-100, 100, J)).to('cuda')
    vsn = rng.uniform(0, 120, J)
    vsn[::5] = 0.0
    vs = torch.as_tensor(vsn).to('cuda')
    idx = torch.zeros(J, dtype=torch.long, device='cuda')
    out = {}
    for fused in (True, False):
        engine.FUSED_OBJECTIVE = fused
        try:
            out[fused] = spec_fit.chisq_jobs(b, idx, vel, par, vs, dict(npoly=5), cfg)
        finally:
            engine.FUSED_OBJECTIVE = True
    (c1, s1), (c0, s0) = out[True], out[False]
    assert torch.equal(s0, s1)
    assert torch.isfinite(c1).all()
    sc = torch.clamp(c0.abs(), min=1e3)
    assert float(((c1 - c0).abs() / sc).max()) < 1e-11


def test_objective_fused_wide_rotational_kernels(cases, config):
    """v sin i of 150-480 km/s: rotational kernels of 6-20 template pixels half width,
    beyond the register-window FIR (8) -- the general FIR (eight outputs per trip, then
    four) of the one-kernel objective against vsini_kernel + the kernel chain"""
    from rvspecfit_amd import engine, spec_fit
    b, _ = spec_fit.as_batch(_sds(cases, 'c1'))
    rng = np.random.RandomState(8)
    J = 64
    par = torch.as_tensor(np.stack([rng.uniform(4600, 7000, J), rng.uniform(1., 4., J),
                                    rng.uniform(-1.8, -0.1, J),
                                    rng.uniform(0.0, 0.4, J)], 1)).to('cuda')
    vel = torch.as_tensor(rng.uniform(-300, 300, J)).to('cuda')
    vs = torch.as_tensor(np.linspace(150., 480., J)).to('cuda')
    idx = torch.zeros(J, dtype=torch.long, device='cuda')
    out = {}
    for fused in (True, False):
        engine.FUSED_OBJECTIVE = fused
        try:
            out[fused] = spec_fit.chisq_jobs(b, idx, vel, par, vs, dict(npoly=10),
                                             config)
        finally:
            engine.FUSED_OBJECTIVE = True
    (c1, s1), (c0, s0) = out[True], out[False]
    assert torch.equal(s0, s1) and int(s1.abs().sum()) == 0
    sc = torch.clamp(c0.abs(), min=1e3)
    assert float(((c1 - c0).abs() / sc).max()) < 1e-11


def test_objective_job_order_and_device_count(cases, config):
    """700 jobs in one rvs_objective_fused launch: (i) from 512 jobs up the blocks
    take the jobs in the order of their grid cell (objective_order_kernel) -- which
    block evaluates a job must not change a bit of its value (option obj_sort = 0: the
    plain order); (ii) with a job count on the device (rvs_objective_fused_n, what
    the lock-step optimiser passes) the first n jobs have the values of the full
    launch and the rows behind them are not written."""
    from rvspecfit_amd import engine, spec_fit, spec_inter
    sds = _sds(cases, 'c3')
    b, _ = spec_fit.as_batch(sds)
    libs = spec_inter.get_libs(b.names, config)
    rng = np.random.RandomState(11)
    J = 700
    par = np.stack([rng.uniform(5000, 6800, J), rng.uniform(1.5, 4.5, J),
                    rng.uniform(-1.5, -0.1, J), rng.uniform(0.0, 0.4, J)], 1)
    par[::37, 0] = 9000.0          # a few points outside the grid
    par = torch.as_tensor(par).to('cuda')
    vel = torch.as_tensor(rng.uniform(-300, 300, J)).to('cuda')
    vs = torch.as_tensor(rng.uniform(0, 120, J)).to('cuda')
    js = torch.zeros(J, dtype=torch.int32, device='cuda')
    kw = dict(npoly=10, rbf=True, job_spec=js)
    with np.errstate(all='ignore'):
        c1, s1 = engine.objective_fused(b, libs, par, vs, vel, **kw)
        with _lib.option('obj_sort', 0):
            c0, s0 = engine.objective_fused(b, libs, par, vs, vel, **kw)
        assert torch.equal(c0, c1) and torch.equal(s0, s1)
        assert torch.isfinite(c1).all()
        for n in (0, 1, 299, 513, 700, 5000):
            cnt = torch.tensor([n], dtype=torch.int32, device='cuda')
            out = torch.full((J, ), -7.0, dtype=torch.float64, device='cuda')
            c2, s2 = engine.objective_fused(b, libs, par, vs, vel, njobs=cnt,
                                            out=out, **kw)
            m = min(n, J)
            assert torch.equal(c2[:m], c1[:m]) and torch.equal(s2[:m], s1[:m])
            assert (c2[m:] == -7.0).all()


@pytest.mark.parametrize('kind', ['triangulation', 'nn'])
def test_objective_from_template(cases, gpu, kind):
    """rvs_objective_from_template (template rows of a Delaunay / MLP evaluator
    from their own kernel, then FIR + spline solve + chi^2 in one kernel) against
    the chain of stand-alone kernels (rvs_vsini_convolve -> rvs_spline_construct
    -> rvs_chisq_point) on the same rows, and -- Delaunay -- against the
    reference's get_chisq"""
    from rvspecfit_amd import engine, spec_fit, spec_inter
    from rvspecfit_amd.library import TemplateLibrary
    rng = np.random.RandomState(12)
    if kind == 'triangulation':
        g = dict(np.load(os.path.join(GOLD, 'tri_cases.npz')))
        cfg = dict(GOLD_CONFIG, template_lib='golden-tri://')
        for n in ('gold_b', 'gold_r'):
            d = np.load(os.path.join(GOLD, 'lib_tri_%s.npz' % n))
            spec_inter.register_library(TemplateLibrary(n, d), 'golden-tri://')
        sds = _sds(cases, 'c1')
        par = np.array([g['c1/t%d/param' % i] for i in range(4)])
        vel = np.array([float(g['c1/t%d/vel' % i]) for i in range(4)])
        vsn = np.array([float(g['c1/t%d/vsini' % i]) for i in range(4)])
        want = np.array([float(g['c1/t%d/value' % i]) for i in range(4)])
        npoly = 10
    else:
        d = dict(np.load(os.path.join(GOLD, 'nn_case.npz')))
        lam = np.exp(np.linspace(np.log(3950.), np.log(5060.),
                                 int(d['dims'][-1])))
        lib = _nn_lib(d, lam)
        lib.name = 'aat_580v'
        spec_inter.register_library(lib, 'golden-nn://')
        cfg = dict(GOLD_CONFIG, template_lib='golden-nn://')
        wave = np.linspace(4000, 5000, 1000)
        err = np.ones(1000) * 0.1
        sds = [spec_fit.SpecData('aat_580v', wave,
                                 rng.normal(wave * 0 + 1, err), err)]
        par = np.stack([rng.uniform(4500, 6500, 6), rng.uniform(1, 4, 6),
                        rng.uniform(-2, 0, 6), rng.uniform(0, 0.4, 6)], axis=1)
        vel = rng.uniform(-300, 300, 6)
        vsn = np.array([np.nan, np.nan, np.nan, 5., 40., 250.])
        want = None
        npoly = 5
    b, _ = spec_fit.as_batch(sds)
    libs = spec_inter.get_libs(b.names, cfg)
    assert engine.can_fuse_objective(b, libs, None, npoly=npoly,
                                     from_template=True)
    assert not engine.can_fuse_objective(b, libs, None, npoly=npoly)
    for with_rot in (False, True):
        ii = np.nonzero(np.isfinite(vsn) == with_rot)[0]
        if not len(ii):
            continue
        tv = torch.as_tensor(vel[ii]).to('cuda')
        tp = torch.as_tensor(par[ii]).to('cuda')
        vs = torch.as_tensor(vsn[ii]).to('cuda') if with_rot else None
        idx = torch.zeros(len(ii), dtype=torch.long, device='cuda')
        out = {}
        for fused in (True, False):
            engine.FUSED_OBJECTIVE = fused
            try:
                with np.errstate(all='ignore'):
                    out[fused] = spec_fit.chisq_jobs(b, idx, tv, tp, vs,
                                                     dict(npoly=npoly), cfg)
            finally:
                engine.FUSED_OBJECTIVE = True
        (c1, s1), (c0, s0) = out[True], out[False]
        assert torch.equal(s0, s1)
        for k, i in enumerate(ii):
            sc = max(abs(c0[k].item()), 1e3)
            assert abs(c1[k].item() - c0[k].item()) < 1e-11 * sc, (kind, i)
            if want is not None:
                assert abs(c1[k].item() - want[i]) < 1e-6 * max(abs(want[i]), 1e3)


def test_resolution_matrix_grid_nd11(cases, config, gold_libs, gold_config):
    """11-diagonal matrices (the DESI width) take the register-window variant of
    the grid kernel; against the oracle's get_chisq on a velocity grid"""
    from rvspecfit_amd import spec_fit, engine
    sds = _sds(cases, 'c1')
    truth = tuple(cases['c1/truth'])
    sds2, osds = [], []
    for sd in sds:
        R = spec_fit.construct_resol_mat(sd.lam, width=0.75)
        taps, nd = engine.resol_taps([R.mat], len(sd.lam))
        assert nd == 11
        sds2.append(spec_fit.SpecData(sd.name, sd.lam, sd.spec, sd.espec,
                                      badmask=sd.badmask, resolution=R))
        osds.append(orc.SpecData(sd.name, sd.lam, sd.spec, sd.espec,
                                 badmask=sd.badmask, resolution=R.mat))
    vg = np.arange(-400., 0., 7.)
    b, _ = spec_fit.as_batch(sds2)
    par = torch.as_tensor(np.array([truth]))[None].to('cuda')
    chisq, st, _ = spec_fit.chisq_grid_jobs(
        b, torch.as_tensor(vg).to('cuda'), par,
        torch.as_tensor([30.], dtype=torch.float64).to('cuda'), dict(npoly=10),
        config)
    got = chisq[0, 0].cpu().numpy()
    want = orc.chisq_grid(osds, vg, [truth], (30., ), dict(npoly=10),
                          gold_config, gold_libs)[:, 0]
    np.testing.assert_allclose(got, want, rtol=1e-8)
    assert int(st.sum().item()) == 0


@pytest.mark.parametrize('linear', [False, True])
@pytest.mark.parametrize('npix,npoly', [(6, 2), (11, 3), (12, 5), (13, 5),
                                        (25, 10), (330, 10), (330, 15)])
def test_resolution_grid_pipe_kernel_vs_point_kernel(cases, config, npix, npoly,
                                                     linear):
    """11-diagonal matrices: the software-pipelined register-window grid kernel
    (whole windows of 12 pixels + a tail, rows one pixel ahead, gathers six) against
    the point kernel, which applies a band of any width from LDS -- arms shorter
    than / equal to / just above one window, several windows, a velocity count that
    is no multiple of 64."""
    from rvspecfit_amd import spec_fit, engine
    sd0 = [x for x in _sds(cases, 'c1') if x.name == 'gold_b'][0]
    rot = 25.
    if linear:   # (no rotation: convolve_vsini needs a log-spaced grid)
        config, rot = _linear_grid_config()[0], None
    i0 = len(sd0.lam) // 4
    i0 = min(i0, len(sd0.lam) - npix)
    lam = sd0.lam[i0:i0 + npix].copy()
    assert len(lam) == npix
    import scipy.sparse
    offs = np.arange(-5, 6)
    rng = np.random.default_rng(npix)
    band = np.exp(-0.5 * (offs[:, None] / 1.7)**2) * (
        1 + 0.05 * rng.standard_normal((11, npix)))
    R = spec_fit.ResolMatrix(scipy.sparse.dia_matrix((band, offs),
                                                     shape=(npix, npix)))
    taps, nd = engine.resol_taps([R.mat], len(lam))
    assert nd == 11
    sds = [spec_fit.SpecData(sd0.name, lam, sd0.spec[i0:i0 + npix].copy(),
                             sd0.espec[i0:i0 + npix].copy(), resolution=R)]
    b, _ = spec_fit.as_batch(sds)
    p = np.array(cases['c1/truth'], dtype=np.float64)
    vg = np.linspace(-310.0, 295.0, 70)
    opt = dict(npoly=npoly)
    g, st, _ = spec_fit.chisq_grid_jobs(
        b, torch.as_tensor(vg).to('cuda'),
        torch.as_tensor(p[None, None, :]).to('cuda'),
        None if rot is None else torch.as_tensor(
            [rot], dtype=torch.float64).to('cuda'), opt, config)
    assert int(st.sum().item()) == 0
    idx = torch.zeros(len(vg), dtype=torch.long, device='cuda')
    c, _ = spec_fit.chisq_jobs(
        b, idx, torch.as_tensor(vg).to('cuda'),
        torch.as_tensor(np.tile(p, (len(vg), 1))).to('cuda'),
        None if rot is None else torch.full(
            (len(vg), ), rot, dtype=torch.float64, device='cuda'), opt, config)
    np.testing.assert_allclose(g.cpu().numpy().reshape(-1), c.cpu().numpy(),
                               rtol=1e-9, atol=1e-7)


def test_pipeline_process_batch(cases, config):
    """fit_batch -> process_batch (the DESI flow of desi_fit.py:288-309) on a
    2-spectrum batch: one fixed-size record per spectrum, consistent with the
    single-spectrum API"""
    from rvspecfit_amd import pipeline, vel_fit
    from rvspecfit_amd.engine import SpecBatch
    lists = [_sds(cases, t) for t in ('c1', 'c3')]
    batch = SpecBatch.from_specdata(lists)
    cfg = dict(config)
    cfg['second_minimizer'] = False
    rec = pipeline.fit_batch(batch, cfg, options=dict(npoly=10))
    pr = pipeline.process_batch(batch, rec, cfg, options=dict(npoly=10))
    assert pr.shape == (2, pipeline.NPROC)
    F = pipeline.PROCESS_FIELDS
    p = pr.cpu().numpy()
    assert np.all(p[:, F.index('minimize_success')] == 1)
    assert abs(p[0, F.index('vel')] - float(cases['c1/vel'])) < 5 * p[0, 1]
    assert abs(p[1, F.index('vel')] - float(cases['c3/vel'])) < 5 * p[1, 1]
    # against the single-spectrum call from the same starting point
    r0 = rec[0].cpu().numpy()
    pd0 = dict(teff=r0[2], logg=r0[3], feh=r0[4], alpha=r0[5],
               vsini=0. if np.isnan(r0[6]) else r0[6])
    one = vel_fit.process(lists[0], pd0, options=dict(npoly=10), config=cfg)
    assert one['vel'] == p[0, F.index('vel')]
    assert one['param']['teff'] == p[0, F.index('teff')]


def test_xcorr_pruned_passes(gpu):
    """pruning the last two FFT passes to the lags that are read back leaves the
    CCF chi^2 surface unchanged up to the compiler's fma-contraction choices
    (DESI size: nfft 8192, n2 = 8^4)"""
    from rvspecfit_amd import engine, synth, spec_inter
    from rvspecfit_amd.library import TemplateLibrary
    import bench
    dev = torch.device('cuda', 0)
    old = bench.ARMS
    bench.ARMS = ('b', )
    try:
        def conv(lam, templ, vsini):
            t = torch.as_tensor(np.ascontiguousarray(templ)).to(dev)
            v = torch.as_tensor(np.ascontiguousarray(vsini)).to(dev)
            return engine.convolve_vsini(lam, t, v).cpu().numpy()
        dicts = bench.build_library_dicts(200, conv)
        cfg = dict(bench.CONFIG)
        cfg['template_lib'] = 'prune-test://'
        for name, d in dicts.items():
            spec_inter.register_library(TemplateLibrary(name, d, device=dev),
                                        'prune-test://')
        tp = bench.truth_params(40, seed=11)
        arms = bench.make_spectra_device(tp, dev)
        out = {}
        for flag in (True, False):
            engine.XCORR_PRUNE = flag
            batch = engine.SpecBatch([engine.ArmData(n, lam, sp, es, bad,
                                                     device=dev)
                                      for n, lam, sp, es, bad in arms])
            libs = spec_inter.get_libs(batch.names, cfg)
            r = engine.ccf_fit(batch, libs, cfg, keep_all=True)
            assert (batch.arms[0].ccf_tables(libs[batch.names[0]], cfg)['prune']
                    is not None) == flag
            out[flag] = (r['all_chisqs'].clone(),
                         r['best_id'].clone(), r['best_vel'].clone(),
                         r['best_ccf'].clone())
    finally:
        engine.XCORR_PRUNE = True
        bench.ARMS = old
    a, b = out[True][0], out[False][0]
    scale = float(b.abs().max())
    print('max |pruned - full| / scale', float((a - b).abs().max()) / scale)
    assert float((a - b).abs().max()) <= 1e-13 * scale
    assert torch.equal(out[True][1], out[False][1])          # best_id
    assert float((out[True][2] - out[False][2]).abs().max()) < 1e-9


@pytest.mark.parametrize('continuum', [1, 0])
@pytest.mark.parametrize('nfft', [64, 128, 256, 512, 1024, 2048, 4096, 8192,
                                  16384])
def test_xcorr_every_plan_vs_numpy(gpu, nfft, continuum):
    """rvs_ccf_xcorr straight through the C-ABI against numpy's rfft / irfft
    (fitter_ccf.py:126-161, 189-216) for every transform size the entry point
    accepts: the pass plans 8,8 / 8,8,2 / 8,8,4 / 8,8,8 / ... with their radix-4
    and radix-2 tails, the twiddle table in LDS at each size, pruned (n2 a power
    of 8) and full last passes, both CCF modes"""
    _xcorr_vs_numpy(nfft, continuum, 3, 5)


def test_xcorr_large_template_set_job_map(gpu):
    """a template set above 16 MB (T = 140 at nfft 8192) takes the XCD-aware,
    grouped (spectrum, template) order of ccf_xcorr_kernel (xc_job): every
    (b, t) row lands where the plain order puts it, padding blocks write
    nothing"""
    with _lib.option('xc_ws', 0):   # (the per-pair kernel)
        _xcorr_vs_numpy(8192, 1, 5, 140)


@pytest.mark.parametrize('continuum', [1, 0])
@pytest.mark.parametrize('nfft,B,T', [(8192, 3, 5), (8192, 5, 140), (8192, 2, 2),
                                      (8192, 1, 77), (4096, 2, 2), (4096, 3, 76),
                                      (4096, 1, 3), (4096, 2, 141)])
def test_xcorr_wave_specialised_equals_per_pair(gpu, nfft, B, T, continuum):
    """ccf_xcorr_ws_kernel (one persistent block per spectrum: producer waves
    keep S*, V* in registers and stream the templates into one LDS image while
    consumer waves transform the other) against ccf_xcorr_kernel (one block per
    (spectrum, template); option xc_ws = 0) and numpy, first call and accumulating
    call, small and large template sets, odd and even T; at nfft 4096 the form that
    takes two templates per iteration.  Same formulas bin by bin and butterfly
    by butterfly; the two kernels are compiled separately, so which product of a
    complex multiplication the compiler fuses into an fma may differ: equal to a
    few ulp of the largest term, not bit for bit.  continuum = 0: the mode without
    continuum normalisation, -c0^2 / c1 at the lags (fitter_ccf.py:204-207) -- one
    correlation per iteration of the persistent block (round 6; the per-pair kernel's
    two passes before)."""
    got_ws = _xcorr_vs_numpy(nfft, continuum, B, T)
    with _lib.option('xc_ws', 0):
        got_pp = _xcorr_vs_numpy(nfft, continuum, B, T)
    np.testing.assert_allclose(got_ws, got_pp, rtol=1e-12,
                               atol=1e-12 * np.abs(got_pp).max())


def _xcorr_vs_numpy(nfft, continuum, B, T):
    from rvspecfit_amd import _lib, ccf_tables
    L = _lib.lib()
    rng = np.random.RandomState(nfft + continuum)
    n2 = nfft // 2
    spec = 1 + 0.2 * rng.standard_normal((B, nfft))
    ivar = rng.uniform(0.5, 2.0, (B, nfft))
    tmod = 1 + 0.3 * rng.standard_normal((T, nfft))
    tfft, tfft2 = np.fft.rfft(tmod, axis=1), np.fft.rfft(tmod**2, axis=1)
    # lags and velocity grid as fitter_ccf builds them (a window of +-7 lags)
    step = 10.0
    maxvel = 6.5 * step
    off = nfft // 2
    vels = -((np.arange(nfft) + off) % nfft - off) * step
    sel = np.abs(vels) < (maxvel + step)
    ind = np.roll(np.nonzero(sel)[0], sel.sum() // 2)[::-1]
    sub = np.ascontiguousarray(vels[ind])
    assert np.all(np.diff(sub) > 0)
    vgrid = np.linspace(-maxvel, maxvel, 41)
    ilo = ccf_tables.interp_tables(sub, vgrid)
    pos = np.array([L.rvs_ccf_fft_pos(nfft, int(n) >> 1) for n in ind])
    lag_pos = (2 * pos + (ind & 1)).astype(np.int32)
    prune = None
    l2 = n2.bit_length() - 1
    if l2 % 3 == 0 and l2 >= 6:
        pm = np.zeros(n2 // 64 + n2 // 8, dtype=np.uint8)
        for p_ in pos:
            pm[n2 // 64 + (int(p_) >> 3)] |= 1 << (int(p_) & 7)
            pm[int(p_) >> 6] |= 1 << ((int(p_) >> 3) & 7)
        prune = torch.as_tensor(pm).to('cuda')
    dev = dict(device='cuda')
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to('cuda')
    twid = np.exp(2j * np.pi * np.arange(n2) / nfft)
    t_spec, t_ivar, t_f, t_f2 = d(spec), d(ivar), d(tfft), d(tfft2)
    t_tw, t_lp, t_lv, t_ilo, t_vg = d(twid), d(lag_pos), d(sub), d(ilo), d(vgrid)
    out = torch.full((B, T, len(vgrid)), 7.0, dtype=torch.float64, **dev)
    work = torch.empty((B, 2, n2 + 1), dtype=torch.complex128, **dev)
    for beta in (0.0, 1.0):     # second call accumulates (the next arm)
        rc = L.rvs_ccf_xcorr(_lib.ptr(t_spec), _lib.ptr(t_ivar), nfft, B,
                             _lib.ptr(t_f), _lib.ptr(t_f2), T, _lib.ptr(t_tw),
                             continuum, _lib.ptr(t_lp), _lib.ptr(t_lv), len(sub),
                             _lib.ptr(t_ilo), _lib.ptr(t_vg), len(vgrid), beta,
                             _lib.ptr(prune), _lib.ptr(out), _lib.ptr(work),
                             _lib.stream())
        assert rc == 0
    got = out.cpu().numpy()
    S = np.conj(np.fft.rfft(spec * ivar, axis=1))
    V = np.conj(np.fft.rfft(ivar, axis=1))
    for b in range(B):
        for t in range(T):
            c0 = np.fft.irfft(tfft[t] * S[b], nfft)[ind]
            c1 = np.fft.irfft(tfft2[t] * V[b], nfft)[ind]
            y = (-2 * c0 + c1) if continuum else (-c0**2 / c1)
            lo = ilo
            ref = (y[lo + 1] - y[lo]) / (sub[lo + 1] - sub[lo]) * \
                (vgrid - sub[lo]) + y[lo]
            np.testing.assert_allclose(got[b, t], 2 * ref, rtol=1e-9,
                                       atol=1e-9 * np.abs(ref).max())
    return got


def test_reference_test_fit_nn_sequence(gpu):
    """tests/test_fit_nn.py of the reference with the NN evaluator (the golden
    network of nn_case.npz): a flat spectrum of pure noise, process with vsini
    fixed and free, first guess.  The reference has no assertions there (and its
    process cannot run beside torch in the build container: no cffi in that
    interpreter), so this pins the path's own contract: the optimiser runs on
    the NN evaluator (torch objective), improves on its starting point, reports
    the chi^2 that get_chisq gives at the point it returns, and a lower-vsini-
    free fit is not worse than the fixed one."""
    from rvspecfit_amd import spec_fit, spec_inter, vel_fit
    d = dict(np.load(os.path.join(GOLD, 'nn_case.npz')))
    lam = np.exp(np.linspace(np.log(3950.), np.log(5060.), int(d['dims'][-1])))
    lib = _nn_lib(d, lam)
    lib.name = 'aat_580v'
    spec_inter.register_library(lib, 'golden-nn://')
    cfg = dict(template_lib='golden-nn://', min_vel=-1000, max_vel=1000,
               min_vel_step=0.2, vel_step0=5, min_vsini=0.1, max_vsini=500,
               second_minimizer=True)
    npix = 1000
    wave = np.linspace(4000, 5000, npix)
    err = np.ones(npix) * 0.1
    dat = np.random.default_rng(400).normal(wave * 0 + 1, err)
    sd = [spec_fit.SpecData('aat_580v', wave, dat, err)]
    opt = {'npoly': 5}
    p0 = {'logg': 2, 'teff': 5000, 'feh': -1, 'alpha': 0.2, 'vsini': 19}
    c0 = spec_fit.get_chisq(sd, 0., (5000., 2., -1., 0.2), rot_params=(19., ),
                            config=cfg, options=opt)
    res = {}
    for tag, fix in (('fixed', ['vsini']), ('free', [])):
        r = vel_fit.process(sd, dict(p0), fixParam=fix, config=cfg, options=opt)
        res[tag] = r
        assert np.isfinite(r['vel']) and np.isfinite(r['chisq'])
        assert r['chisq'] <= c0 + 1e-6
        vs = (19., ) if tag == 'fixed' else (r['vsini'], )
        par = tuple(r['param'][k] for k in ('teff', 'logg', 'feh', 'alpha'))
        again = spec_fit.get_chisq(sd, r['vel'], par, rot_params=vs, config=cfg,
                                   options=opt, full_output=True)
        assert abs(again['chisq'] - r['chisq']) <= 1e-8 * abs(r['chisq'])
        assert r['npix_array'] == [npix] and len(r['yfit'][0]) == npix
        assert abs(np.sum(((dat - r['yfit'][0]) / err)**2) -
                   r['chisq_array'][0]) < 1e-6 * r['chisq_array'][0]
    assert res['free']['chisq'] <= res['fixed']['chisq'] + 0.05
    g = vel_fit.firstguess(sd, config=cfg)
    assert set(g) >= {'teff', 'logg', 'feh', 'alpha'}


def _process_with_torch_machine(batch, p0, cfg, opt, names, fit_vsini=True):
    """vel_fit.process with tests/refmachines/neldermead_torch.minimize over
    vel_fit._Objective in the place of the rvs_nm_* kernels (both Nelder-Mead
    runs of vel_fit.py:624-649)"""
    from refmachines import neldermead_torch
    from rvspecfit_amd import optimizer, vel_fit

    class TorchNM(optimizer.DeviceNelderMead):
        def minimize(self, pobj, simplex, **kw):
            b = pobj.batch
            pd = {k: v for k, v in zip(names, pobj.fixed.T)}
            if pobj.has_vsini:
                pd['vsini'] = torch.zeros(b.S, dtype=torch.float64,
                                          device=b.device)
            mapper = vel_fit.ParamMapper(
                names, pd, [], vel_fit.VSiniMapper(cfg['max_vsini']),
                fitVsini=fit_vsini)
            obj = vel_fit._Objective(b, mapper, cfg, opt, None)
            obj.safe_params = pobj.safe
            return neldermead_torch.minimize(obj, simplex, **kw)
    keep = optimizer.DeviceNelderMead
    optimizer.DeviceNelderMead = TorchNM
    try:
        return vel_fit.process(batch, dict(p0), config=cfg, options=opt)
    finally:
        optimizer.DeviceNelderMead = keep


@pytest.mark.parametrize('maxiter', [10000, 40])
def test_process_nn_library_device_neldermead(gpu, maxiter):
    """BASELINE configs[3]'s evaluator under the optimiser: the simplices of an NN
    library advance on the device (rvs_nm_* state machine; rvs_template_nn for
    the MLP rows of a round, then vsini / spline / rvs_chisq_point) exactly as
    the torch reference machine drives them over the batched get_chisq -- same
    iteration counts, same end points.  maxiter 40: no simplex converges, so
    every spectrum also goes through the restart from its final simplex
    (vel_fit.py:640-649), on the device as well."""
    from rvspecfit_amd import spec_fit, spec_inter, vel_fit
    from rvspecfit_amd.engine import SpecBatch
    d = dict(np.load(os.path.join(GOLD, 'nn_case.npz')))
    lam = np.exp(np.linspace(np.log(3950.), np.log(5060.), int(d['dims'][-1])))
    lib = _nn_lib(d, lam)
    lib.name = 'aat_580v'
    spec_inter.register_library(lib, 'golden-nn://')
    cfg = dict(template_lib='golden-nn://', min_vel=-1000, max_vel=1000,
               min_vel_step=0.2, vel_step0=5, min_vsini=0.1, max_vsini=500,
               second_minimizer=False)
    S, npix = 6, 1000
    wave = np.linspace(4000, 5000, npix)
    rng = np.random.default_rng(41)
    err = np.ones(npix) * 0.05
    tsp = lib.eval_batch(torch.as_tensor([[5200., 2.5, -1., 0.2]]).to('cuda')
                         )[0][0].cpu().numpy()
    base = np.interp(wave, lam, tsp)
    base = base / np.median(base)
    lists = [[spec_fit.SpecData('aat_580v', wave,
                                rng.normal(base, err), err)] for _ in range(S)]
    batch = SpecBatch.from_specdata(lists)
    p0 = dict(teff=rng.uniform(4800, 5600, S), logg=rng.uniform(2, 3, S),
              feh=rng.uniform(-1.4, -0.6, S), alpha=rng.uniform(0.1, 0.3, S),
              vsini=rng.uniform(5, 30, S))
    opt = dict(npoly=5)
    names = ['teff', 'logg', 'feh', 'alpha']
    keep = vel_fit.NM_MAXITER
    vel_fit.NM_MAXITER = maxiter
    try:
        a = vel_fit.process(batch, dict(p0), config=cfg, options=opt)
        b = _process_with_torch_machine(batch, p0, cfg, opt, names)
    finally:
        vel_fit.NM_MAXITER = keep
    if maxiter < 100:
        assert not bool(a['minimize_success'].any())
        assert int(a['nm_nit'].min()) == 2 * maxiter
    else:
        assert bool(a['minimize_success'].all())
    assert torch.equal(a['nm_nit'], b['nm_nit'])
    assert torch.equal(a['nm_nfev'], b['nm_nfev'])
    assert torch.equal(a['minimize_success'], b['minimize_success'])
    assert torch.equal(a['nm_vel'], b['nm_vel'])
    for k in names:
        assert torch.equal(a['param'][k], b['param'][k]), k
    assert torch.equal(a['chisq'], b['chisq'])


def test_ccf_readback_paths_agree(cases, config):
    """the cross-correlation kernel has two read-back paths: tables prefetched
    into registers (<= 512 lags and velocities, the normal case) and the loop
    (more).  A 2 km/s grid (1001 velocities) takes the loop; where the two grids
    share a velocity the interpolated CCF must be the same number."""
    from rvspecfit_amd import spec_fit, fitter_ccf
    sds = gold_specdata(cases, 'c1', spec_fit.SpecData)
    out = {}
    for step in (5, 2):
        cfg = dict(config)
        cfg['vel_step0'] = step
        r = fitter_ccf.fit(sds, cfg)
        out[step] = (np.round(np.asarray(r['vel_grid']), 6),
                     np.asarray(r['best_ccf']))
    assert len(out[5][0]) <= 512 < len(out[2][0])
    common = np.intersect1d(out[5][0], out[2][0])
    assert len(common) > 100
    a = out[5][1][np.searchsorted(out[5][0], common)]
    b = out[2][1][np.searchsorted(out[2][0], common)]
    np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
def test_grid_redo_third_tier_singular_matrix(cases, config):
    """The tiers behind the velocity-grid kernel (engine.chisq_grid.by_point_kernel,
    spec_fit.py:337-354): a job whose normal matrix NEITHER kernel can factor -- a
    template that vanishes everywhere makes it exactly zero -- goes through
    rvs_chisq_full's eigen branch; the job is flagged, its value is what
    rvs_chisq_full returns for it, and the other jobs of the call are untouched."""
    from rvspecfit_amd import _lib, engine, spec_fit, spec_inter
    sds = _sds(cases, 'c1')
    b, _ = spec_fit.as_batch(sds)
    libs = spec_inter.get_libs(b.names, config)
    vg = torch.as_tensor(np.linspace(-100., 100., 21)).to('cuda')
    par = torch.as_tensor(np.array([cases['c1/g3/params_list'][0]] * 2)).to('cuda')
    coefs, outs = [], []
    for arm in b.arms:
        c, o = engine.build_templates(libs[arm.name], par, None)
        coefs.append(c)
        outs.append(o)
    js = torch.zeros(2, dtype=torch.int32, device='cuda')
    jt = torch.arange(2, dtype=torch.int32, device='cuda')
    ref, st0 = engine.chisq_grid(b, libs, coefs, outs, vg, npoly=10, job_spec=js,
                                 job_templ=jt)
    assert int(st0.sum().item()) == 0
    dead = [c.clone() for c in coefs]
    for c in dead:
        c[1] = 0.0                      # template 1: zero spline records
    got, st = engine.chisq_grid(b, libs, dead, outs, vg, npoly=10, job_spec=js,
                                job_templ=jt)
    st = st.cpu().numpy()
    assert st[0] == 0 and np.array_equal(got[0].cpu().numpy(), ref[0].cpu().numpy())
    assert st[1] & _lib.ST_CHOL_FALLBACK and st[1] & _lib.ST_NONFINITE
    full = engine.chisq_full(b, libs, dead, vg[:1].expand(1).contiguous(), npoly=10,
                             job_spec=js[1:], job_templ=jt[1:], want_models=False)
    # the eigen branch on an exactly singular matrix: not a finite likelihood
    assert all(f['status'][0].item() & _lib.ST_CHOL_FALLBACK for f in full)
    assert not np.isfinite(got[1].cpu().numpy()).any()
