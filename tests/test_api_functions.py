"""The reference's public helper functions on the hot path, called one by one --
the names a user's own likelihood code imports next to get_chisq (SURVEY 8 rows A1,
A2, A3, A6, A10, A14, A15) -- against vectors of the reference itself
(tests/golden/api_cases.npz, written by make_golden_api.py importing it).

CPU half: the oracle's restatements against those vectors.  GPU half: the product's
functions of the same names (rvspecfit_amd.spec_fit / spec_inter / make_ccf /
fitter_ccf), which run on the kernels.
"""
import itertools
import os

import numpy as np
import pytest

from conftest import GOLD, gold_lib_dict
from oracle import rvs_oracle as orc

BASES = ((1, True), (3, True), (10, True), (16, True), (7, False), (15, False))


@pytest.fixture(scope='module')
def api():
    return dict(np.load(os.path.join(GOLD, 'api_cases.npz')))


# --------------------------------------------------------------------------
# CPU: the oracle against the reference's vectors
# --------------------------------------------------------------------------
def test_oracle_bases_vs_reference(api):
    for npoly, rbf in BASES:
        np.testing.assert_allclose(
            orc.get_poly_basis(api['basis/lam'], npoly, rbf),
            api['basis/p%d_%d' % (npoly, rbf)], rtol=1e-13, atol=1e-15)


def test_oracle_chisq0_vs_reference(api):
    a = api
    args = (a['chisq0/spec'], a['chisq0/templ'], a['chisq0/polys'])
    assert abs(orc.get_chisq0(*args, espec=a['chisq0/espec']) /
               a['chisq0/value'] - 1) < 1e-12
    c, co = orc.get_chisq0(*args, get_coeffs=True, espec=a['chisq0/espec'])
    assert abs(c / a['chisq0/value_coeffs'] - 1) < 1e-12
    np.testing.assert_allclose(co, a['chisq0/coeffs'], rtol=1e-8)
    e = a['chisq0/espec']
    c = orc.get_chisq0(args[0] / e, args[1] / e, args[2])
    assert abs(c / a['chisq0/value_noespec'] - 1) < 1e-12


def test_oracle_vsini_kernels_vs_reference(api):
    for i, R in enumerate(api['vsini/R']):
        k = orc.compute_vsini_kernel(R)
        assert k.shape == api['vsini/k%d' % i].shape
        # (the far taps of a wide kernel are differences of primitives close to each
        # other: absolute 1e-14 of a kernel that sums to 1)
        np.testing.assert_allclose(k, api['vsini/k%d' % i], rtol=1e-12, atol=5e-14)
    np.testing.assert_allclose(orc.compute_vsini_kernel(3.3, eps=0.3),
                               api['vsini/k_eps03'], rtol=1e-12)


def _ccfconf(api, cont):
    d = dict(logl0=np.log(3990.), logl1=np.log(5010.), npoints=2048, continuum=cont,
             maxcontpts=20)
    if cont:
        d['splinestep'] = float(api['ccf/conf_splinestep'])
    return d


CCF_CASES = [('cont', True, False, 10), ('cont_mask', True, True, 10),
             ('nocont', False, False, 10), ('nocont_mask', False, True, 10),
             ('cont_maxerr3', True, False, 3)]


@pytest.mark.parametrize('tag,cont,masked,maxerr', CCF_CASES)
def test_oracle_preprocess_vs_reference(api, tag, cont, masked, maxerr):
    r1, r2 = orc.preprocess_data(api['ccf/lam'], api['ccf/spec'], api['ccf/espec'],
                                 _ccfconf(api, cont),
                                 badmask=api['ccf/badmask'] if masked else None,
                                 maxerr=maxerr)
    np.testing.assert_array_equal(r2 == 0, api['ccf/%s/ivar' % tag] == 0)
    np.testing.assert_allclose(r1, api['ccf/%s/spec' % tag], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(r2, api['ccf/%s/ivar' % tag], rtol=1e-6)


def test_specdata_fields_are_read_only():
    """spec_fit.SpecData (spec_fit.py:70-145): properties over a frozen record"""
    from rvspecfit_amd import spec_fit
    lam = np.linspace(4000, 4010, 11)
    sd = spec_fit.SpecData('arm', lam, np.ones(11), np.full(11, 0.5))
    assert sd.name == 'arm' and sd.resolution is None
    np.testing.assert_array_equal(sd.spec_error_ratio, np.full(11, 2.0))
    assert sd.badmask.dtype == bool and not sd.badmask.any()
    assert sd.lam.dtype == np.float64 and sd.lam.flags.c_contiguous
    for f in ('name', 'lam', 'spec', 'espec', 'badmask', 'resolution',
              'spec_error_ratio'):
        with pytest.raises(AttributeError):
            setattr(sd, f, 1)
    assert hash(sd) == hash(sd.objid)
    sd32 = spec_fit.SpecData('arm', lam, np.ones(11), np.ones(11), dtype=np.float32)
    assert sd32.spec.dtype == np.float32


def test_public_names_of_the_path_exist():
    """every public name of the reference's hot-path modules that SURVEY 8(a) lists"""
    from rvspecfit_amd import spec_fit, spec_inter, fitter_ccf, make_ccf, vel_fit
    want = {
        spec_fit: ['SpecData', 'ResolMatrix', 'LRUDict', 'get_poly_basis',
                   'get_basis', 'get_chisq0', 'getCurTempl', 'construct_resol_mat',
                   'convolve_resol', 'compute_vsini_kernel', 'convolve_vsini',
                   'getRVInterpol', 'evalRV', 'param_dict_to_tuple',
                   'get_chisq_continuum', 'get_chisq', 'find_best'],
        spec_inter: ['TriInterp', 'GridOutsideCheck', 'GridInterp',
                     'SpecInterpolator', 'interp_cache', 'getInterpolator',
                     'getSpecParams'],
        fitter_ccf: ['CCFCache', 'get_ccf_info', 'fit'],
        make_ccf: ['get_continuum_prefix', 'get_ccf_info_name', 'get_ccf_dat_name',
                   'get_ccf_mod_name', 'get_ccf_config', 'preprocess_data',
                   'interp_masker', 'to_power_two'],
        vel_fit: ['firstguess', 'process', 'VSiniMapper', 'ParamMapper',
                  'get_hess_inv', 'chisq_func0', 'chisq_func', 'hess_func'],
    }
    for mod, names in want.items():
        for n in names:
            assert hasattr(mod, n), (mod.__name__, n)


def test_ccf_config_and_file_names(api):
    from rvspecfit_amd import make_ccf
    c = make_ccf.get_ccf_config(logl0=np.log(3990.), logl1=np.log(5010.),
                                npoints=2048, splinestep=1000, maxcontpts=20)
    assert c['continuum'] and c['splinestep'] == float(api['ccf/conf_splinestep'])
    c = make_ccf.get_ccf_config(logl0=np.log(3990.), logl1=np.log(5010.),
                                npoints=2048, splinestep=200, maxcontpts=8)
    assert c['splinestep'] == float(api['ccf/conf_wide_splinestep'])
    c = make_ccf.get_ccf_config(logl0=1., logl1=2., npoints=64, splinestep=None)
    assert not c['continuum'] and 'splinestep' not in c
    assert make_ccf.get_ccf_info_name('b') == 'ccf_b.h5'
    assert make_ccf.get_ccf_dat_name('b', False) == 'ccfdat_nocont_b.npz'
    assert make_ccf.get_ccf_mod_name('b', True) == 'ccfmod_b.npy'


# --------------------------------------------------------------------------
# GPU: the product's functions of the same names
# --------------------------------------------------------------------------
@pytest.mark.gpu
def test_get_poly_basis_and_get_basis(api):
    from rvspecfit_amd import spec_fit
    lam = api['basis/lam']
    for npoly, rbf in BASES:
        b = spec_fit.get_poly_basis(lam, npoly, rbf=rbf)
        assert b.shape == (npoly, len(lam)) and b.flags.c_contiguous
        np.testing.assert_allclose(b, api['basis/p%d_%d' % (npoly, rbf)],
                                   rtol=1e-12, atol=1e-14)
    sd = spec_fit.SpecData('x', lam, np.ones_like(lam), np.ones_like(lam))
    b1 = spec_fit.get_basis(sd, 10)
    assert spec_fit.get_basis(sd, 10) is b1          # cached per dataset
    np.testing.assert_allclose(b1, api['basis/p10_1'], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(spec_fit.get_basis(sd, 7, rbf=False),
                               api['basis/p7_0'], rtol=1e-12, atol=1e-14)


@pytest.mark.gpu
def test_get_chisq0(api):
    import torch
    from rvspecfit_amd import spec_fit
    a = api
    spec, templ, polys, e = (a['chisq0/spec'], a['chisq0/templ'],
                             a['chisq0/polys'], a['chisq0/espec'])
    c = spec_fit.get_chisq0(spec, templ, polys, espec=e)
    assert isinstance(c, float)
    assert abs(c / float(a['chisq0/value']) - 1) < 1e-11
    c, co = spec_fit.get_chisq0(spec, templ, polys, get_coeffs=True, espec=e)
    assert abs(c / float(a['chisq0/value_coeffs']) - 1) < 1e-11
    np.testing.assert_allclose(co, a['chisq0/coeffs'], rtol=1e-8)
    c = spec_fit.get_chisq0(spec / e, templ / e, polys)
    assert abs(c / float(a['chisq0/value_noespec']) - 1) < 1e-11
    # a matrix that does not factor: the eigen tier (the reference's SVD)
    c, co = spec_fit.get_chisq0(spec, templ, a['chisq0/polys_singular'],
                                get_coeffs=True, espec=e)
    # (nothing here is reproducible: the value carries the logarithm of a singular
    # value that is rounding noise -- 5e-12 of 8e5 -- and the coefficients come out of
    # a cancellation of terms of 1e11: the reference's own are multiples of 1/16
    # (1.3125, -1.0625), numpy 2.2's LAPACK gives 0.717, -1.717 and -2 log L =
    # +348148 where the reference has -3580.  What holds: the call returns, finite.)
    assert np.isfinite(c) and np.isfinite(co).all()
    # rows of a batch: each its own fit
    S = 5
    rng = np.random.default_rng(3)
    sp2 = spec[None, :] + e[None, :] * rng.normal(size=(S, len(spec)))
    ch, cf = spec_fit.get_chisq0(sp2, templ, polys, get_coeffs=True,
                                 espec=np.tile(e, (S, 1)))
    assert ch.shape == (S, ) and cf.shape == (S, polys.shape[0]) and ch.is_cuda
    for i in range(S):
        w, wc = orc.get_chisq0(sp2[i], templ, polys, get_coeffs=True, espec=e)
        assert abs(ch[i].item() / w - 1) < 1e-11
        np.testing.assert_allclose(cf[i].cpu().numpy(), wc, rtol=1e-8)
    # device tensors in
    ch2 = spec_fit.get_chisq0(torch.as_tensor(sp2).cuda(), torch.as_tensor(
        templ).cuda(), torch.as_tensor(polys).cuda(),
        espec=torch.as_tensor(np.tile(e, (S, 1))).cuda())
    assert torch.equal(ch, ch2)


@pytest.mark.gpu
def test_compute_vsini_kernel(api):
    from rvspecfit_amd import spec_fit
    for i, R in enumerate(api['vsini/R']):
        k = spec_fit.compute_vsini_kernel(float(R))
        want = api['vsini/k%d' % i]
        assert k.shape == want.shape
        np.testing.assert_allclose(k, want, rtol=1e-11, atol=1e-13)
        assert abs(k.sum() - 1) < 1e-14
    np.testing.assert_allclose(spec_fit.compute_vsini_kernel(3.3, eps=0.3),
                               api['vsini/k_eps03'], rtol=1e-11, atol=1e-13)
    with pytest.raises(AssertionError):
        spec_fit.compute_vsini_kernel(0.0)


@pytest.mark.gpu
@pytest.mark.parametrize('tag,cont,masked,maxerr', CCF_CASES)
def test_make_ccf_preprocess_data(api, tag, cont, masked, maxerr):
    from rvspecfit_amd import make_ccf
    conf = make_ccf.get_ccf_config(logl0=np.log(3990.), logl1=np.log(5010.),
                                   npoints=2048,
                                   splinestep=1000 if cont else None)
    r1, r2 = make_ccf.preprocess_data(api['ccf/lam'], api['ccf/spec'],
                                      api['ccf/espec'], ccfconf=conf,
                                      badmask=api['ccf/badmask'] if masked else None,
                                      maxerr=maxerr)
    w1, w2 = api['ccf/%s/spec' % tag], api['ccf/%s/ivar' % tag]
    assert r1.shape == w1.shape == (2048, )
    # masks are integer work: the zero pattern is the reference's
    np.testing.assert_array_equal(r2 == 0, w2 == 0)
    np.testing.assert_array_equal(r1 == 0, w1 == 0)
    # continuum: the device Levenberg-Marquardt and scipy's TRF stop at different
    # points of a 1e-8 fit
    tol = 2e-5 if cont else 1e-10   # (fused multiply-adds in the rebinning)
    np.testing.assert_allclose(r1, w1, rtol=tol, atol=tol * 1e-2)
    np.testing.assert_allclose(r2, w2, rtol=tol)


@pytest.mark.gpu
def test_make_ccf_preprocess_data_rows(api):
    """[S, npix] in: rows are the single calls"""
    from rvspecfit_amd import make_ccf
    conf = make_ccf.get_ccf_config(logl0=np.log(3990.), logl1=np.log(5010.),
                                   npoints=2048, splinestep=1000)
    rng = np.random.default_rng(5)
    sp = api['ccf/spec'][None, :] * rng.uniform(0.5, 2, size=(3, 1))
    es = np.tile(api['ccf/espec'], (3, 1))
    R1, R2 = make_ccf.preprocess_data(api['ccf/lam'], sp, es, ccfconf=conf)
    assert R1.shape == (3, 2048) and R1.is_cuda
    for i in range(3):
        r1, r2 = make_ccf.preprocess_data(api['ccf/lam'], sp[i], es[i], ccfconf=conf)
        np.testing.assert_array_equal(R1[i].cpu().numpy(), r1)
        np.testing.assert_array_equal(R2[i].cpu().numpy(), r2)


def _grid(api):
    uvecs = [api['grid/uvec%d' % i] for i in range(3)]
    return uvecs, api['grid/idgrid'], api['grid/vecs'], api['grid/dats'], api['grid/P']


@pytest.mark.gpu
def test_grid_interp_and_outside_check(api):
    from rvspecfit_amd import spec_inter
    from test_numpy_expf import host_numpy_expf_is_published_algorithm
    uvecs, idgrid, vecs, dats, P = _grid(api)
    for exp in (True, False):
        GI = spec_inter.GridInterp(uvecs, idgrid, vecs, dats, exp=exp)
        want = api['grid/spec_exp%d' % exp]
        got = np.array([GI(p) for p in P])
        # (a nearest-neighbour row is exp of a float32 in float32, as numpy does it)
        np.testing.assert_allclose(got, want, rtol=1e-12 if not exp else 2e-7)
        inside = api['grid/outside'] == 0
        np.testing.assert_allclose(got[inside], want[inside], rtol=1e-12)
        if host_numpy_expf_is_published_algorithm():
            np.testing.assert_array_equal(got[~inside], want[~inside])
        np.testing.assert_array_equal(GI.batch(P).cpu().numpy(), got)
    np.testing.assert_array_equal([GI.get_nearest(p) for p in P],
                                  api['grid/nearest'])
    GO = spec_inter.GridOutsideCheck(uvecs, vecs, idgrid)
    out = np.array([float(GO(p)) for p in P])
    np.testing.assert_array_equal(out == 0, api['grid/outside'] == 0)
    np.testing.assert_allclose(out, api['grid/outside'], rtol=1e-13)
    assert (~(api['grid/outside'] == 0)).sum() >= 5   # both kinds of outside are in P
    with pytest.raises(TypeError):
        spec_inter.GridInterp(uvecs, idgrid, vecs, dats.astype(np.float64))


@pytest.mark.gpu
def test_tri_interp(api):
    import scipy.spatial
    from rvspecfit_amd import spec_inter
    tri = scipy.spatial.Delaunay(api['tri/points'])
    # (Qhull is deterministic for the same points: the vectors' own triangulation)
    np.testing.assert_array_equal(tri.simplices, api['tri/simplices'])
    for exp in (True, False):
        TI = spec_inter.TriInterp(tri, api['tri/dats'], exp=exp)
        want = api['tri/spec_exp%d' % exp]
        for p, w in zip(api['tri/P'], want):
            r = TI(p)
            if np.isnan(w).all():
                assert np.ndim(r) == 0 and np.isnan(r)
            else:
                np.testing.assert_allclose(r, w, rtol=1e-11, atol=1e-14)
    assert np.isnan(want).all(axis=1).sum() == 2


@pytest.mark.gpu
def test_get_ccf_info():
    from rvspecfit_amd import fitter_ccf, spec_inter
    from rvspecfit_amd.library import TemplateLibrary
    from conftest import GOLD_CONFIG
    cfg = dict(GOLD_CONFIG, template_lib='golden-api://')
    d = gold_lib_dict('gold_b')
    spec_inter.register_library(TemplateLibrary('gold_b', d), 'golden-api://')
    fitter_ccf.CCFCache.ccfs.pop('gold_b', None)
    fft, fft2, mod, info = fitter_ccf.get_ccf_info('gold_b', cfg)
    np.testing.assert_array_equal(fft, d['ccf_fft'])
    np.testing.assert_array_equal(fft2, d['ccf_fft2'])
    assert fft.dtype == np.complex128
    np.testing.assert_array_equal(info['params'], d['ccf_params'])
    np.testing.assert_array_equal(info['vsinis'], d['ccf_vsinis'])
    assert info['parnames'] == tuple(str(_) for _ in d['parnames'])
    cc = info['ccfconf']
    assert cc['npoints'] == int(d['ccf_npoints']) and cc['continuum'] is True
    assert cc['logl0'] == float(d['ccf_logl0'])
    if mod is not None:
        np.testing.assert_array_equal(mod, d['ccf_mod'])
    assert 'gold_b' in fitter_ccf.CCFCache.ccfs
    assert fitter_ccf.get_ccf_info('gold_b', cfg)[0] is fft   # cached


# --------------------------------------------------------------------------
# round 6: the objective functions of vel_fit.process and make_ccf's gap filler by
# name (api2_cases.npz, make_golden_api2.py)
# --------------------------------------------------------------------------
def test_interp_masker_and_to_power_two():
    """make_ccf.interp_masker (make_ccf.py:288-327) against the reference's outputs:
    interior gaps, both edges, everything masked, nothing masked"""
    from rvspecfit_amd import make_ccf
    g = np.load(os.path.join(GOLD, 'api2_cases.npz'))
    for m, want in zip(g['im/masks'], g['im/out']):
        got = make_ccf.interp_masker(g['im/lam'], g['im/spec'], m)
        np.testing.assert_allclose(got, want, rtol=1e-14, atol=0)
    assert [make_ccf.to_power_two(_) for _ in (1, 2, 3, 1000, 4096, 4097)] == \
        [1, 2, 4, 1024, 4096, 8192]


@pytest.mark.gpu
@pytest.mark.parametrize('t', ['f0', 'f1'])
def test_chisq_func_by_name(t):
    """vel_fit.chisq_func0 / chisq_func / hess_func (vel_fit.py:210-269) with the
    reference's own `args` dictionary and its one-vector ParamMapper.forward: priors, a
    fixed parameter, vsini inside and outside its range (the penalty), a velocity
    outside [min_vel, max_vel] and a non-finite parameter (1e30 without evaluating)"""
    from rvspecfit_amd import spec_fit, spec_inter, vel_fit
    from rvspecfit_amd.library import TemplateLibrary
    from conftest import GOLD_CONFIG
    g = np.load(os.path.join(GOLD, 'api2_cases.npz'))
    cases = np.load(os.path.join(GOLD, 'cases.npz'))
    cfg = dict(GOLD_CONFIG, template_lib='golden-api2://')
    for n in ('gold_b', 'gold_r'):
        spec_inter.register_library(TemplateLibrary(n, gold_lib_dict(n)),
                                    cfg['template_lib'])
    case = str(g[t + '/case'])
    names = [str(_) for _ in cases[case + '/names']]
    sds = [spec_fit.SpecData(n, cases['%s/%s/lam' % (case, n)],
                             cases['%s/%s/spec' % (case, n)],
                             cases['%s/%s/espec' % (case, n)],
                             badmask=cases['%s/%s/badmask' % (case, n)]) for n in names]
    pd0 = dict(zip([str(_) for _ in g[t + '/pd0_keys']],
                   [float(_) for _ in g[t + '/pd0_vals']]))
    fix = [str(_) for _ in g[t + '/fix']]
    pri = None
    if t + '/prior_keys' in g:
        pri = {str(k): tuple(v) for k, v in zip(g[t + '/prior_keys'],
                                                g[t + '/prior_vals'])}
    fit_vsini = 'vsini' in pd0 and 'vsini' not in fix
    vm = vel_fit.VSiniMapper(cfg['max_vsini']) if fit_vsini else None
    pm = vel_fit.ParamMapper(['teff', 'logg', 'feh', 'alpha'], pd0, fix, vm,
                             fitVsini=fit_vsini)
    args = dict(specdata=sds, paramMapper=pm, resolParams=None, options=dict(npoly=10),
                config=cfg, priors=pri, min_vel=cfg['min_vel'],
                max_vel=cfg['max_vel'])
    npix = sum(len(_.lam) for _ in sds)

    def close(a, b):
        return abs(a - b) <= 1e-6 * max(abs(b), npix)
    for i, p in enumerate(g[t + '/ps']):
        want = float(g[t + '/chisq_func'][i])
        with np.errstate(all='ignore'):
            got = vel_fit.chisq_func(np.array(p), args)
        assert got == want if want == 1e30 else close(got, want), (i, got, want)
        if want == 1e30:
            continue
        pd = pm.forward(np.array(p))
        assert isinstance(pd['params'], list) and pd['penalty'] >= 0
        w0 = g[t + '/chisq_func0'][i]
        assert close(vel_fit.chisq_func0(pd, args), w0[0])
        assert close(vel_fit.chisq_func0(pd, args, outside_penalty=False), w0[1])
        pd['params'] = np.array(pd['params'], dtype=float)
        h = vel_fit.hess_func(pd['params'] * np.array([1.001, 1.0, 1.0, 1.0]), pd, args)
        assert close(2 * h, 2 * float(g[t + '/hess_func'][i]))
