"""Pin the CPU oracle against vectors captured from the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import scipy.interpolate

from oracle import rvs_oracle as orc
from conftest import gold_specdata, GOLD
import os

TAGS = ['c0', 'c1', 'c2', 'c3']


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))


@pytest.mark.parametrize('kind', ['log', 'lin'])
def test_spline_vs_reference_golden(cases, kind):
    g = lambda k: cases['spline/%s/%s' % (kind, k)]
    S = orc.Spline(g('xs'), g('ys'), log_step=(kind == 'log'))
    for k in 'ABCDh':
        np.testing.assert_allclose(getattr(S, k), g(k), rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(S(g('evalx')), g('ret'), rtol=1e-12, atol=1e-13)


def test_spline_vs_reference_c_source():
    ref = orc.load_reference_spliner()
    if ref is None:
        pytest.skip('oracle/_ref not built (reference absent)')
    rng = np.random.RandomState(5)
    xs = np.exp(np.linspace(np.log(3500.), np.log(5900.), 6215))
    ys = 1 + 0.3 * rng.standard_normal(len(xs))
    ex = np.sort(rng.uniform(3600, 5800, size=2751))
    a, b = orc.Spline(xs, ys), orc.Spline(xs, ys, lib=ref)
    for k in 'ABCDh':
        np.testing.assert_allclose(getattr(a, k), getattr(b, k), rtol=1e-11,
                                   atol=1e-12)
    np.testing.assert_allclose(a(ex), b(ex), rtol=1e-12, atol=1e-13)


def test_spline_vs_scipy_natural():
    # the reference's own pin: tests/test_spline.py:6-21
    rng = np.random.RandomState(1)
    x = np.linspace(1000, 2000, 1000)
    y = 0.00001 * x**2 + rng.normal(size=len(x))
    xn = rng.uniform(1000, 2000, size=10000)
    xn = np.sort(xn[xn < 1999.9999])
    yref = scipy.interpolate.CubicSpline(x, y, bc_type='natural')(xn)
    assert np.allclose(yref, orc.Spline(x, y, log_step=False)(xn))
    x = 10**np.linspace(3, 4, 1000)
    y = np.sin(x / 10) + rng.normal(size=len(x))
    xn = np.sort(rng.uniform(1000, 2000, size=10000))
    yref = scipy.interpolate.CubicSpline(x, y, bc_type='natural')(xn)
    assert np.allclose(yref, orc.Spline(x, y, log_step=True)(xn))


def test_spline_error_codes():
    xs = np.exp(np.linspace(1, 2, 50))
    S = orc.Spline(xs, np.ones(50))
    with pytest.raises(AssertionError):
        S(np.array([xs[0] * 0.9, xs[3]]))
    with pytest.raises(AssertionError):
        S(np.array([xs[3], xs[-1]]))
    xs2 = xs.copy()
    xs2[2] *= 1.001
    with pytest.raises(AssertionError):
        orc.Spline(xs2, np.ones(50))(np.array([xs[5]]))


@pytest.mark.parametrize('name', ['gold_b', 'gold_r'])
def test_polylinear_and_outside(cases, gold_libs, name):
    lib = gold_libs[name]
    P = cases['interp/params']
    for i, p in enumerate(P):
        spec, info = lib.eval(p, details=True)
        np.testing.assert_allclose(spec, cases['interp/%s/eval' % name][i],
                                   rtol=2e-7)  # float32 dats, order of sums
        o = lib.outside_flag(p)
        oref = cases['interp/%s/outside' % name][i]
        if np.isfinite(oref):
            assert abs(o - oref) <= 1e-12 * max(1, abs(oref))
        else:
            assert not np.isfinite(o)
        if info['nearest'] >= 0:
            assert info['nearest'] == cases['interp/%s/nearest' % name][i]


def test_vsini_kernel_and_convolution(cases, gold_libs):
    for i, R in enumerate(cases['vsini/R']):
        np.testing.assert_allclose(orc.compute_vsini_kernel(R),
                                   cases['vsini/kernel_%d' % i], rtol=1e-10,
                                   atol=1e-13)
    lib = gold_libs['gold_b']
    for i, v in enumerate(cases['vsini/vsinis']):
        np.testing.assert_allclose(
            orc.convolve_vsini(lib.lam, cases['vsini/templ'], v),
            cases['vsini/conv_%d' % i], rtol=1e-12)


def test_cur_templ(cases, gold_libs):
    P = cases['interp/params']
    rots = [None, (10., ), (300., )]
    for name, lib in gold_libs.items():
        for ip in (0, 3, 4):
            for ir, rot in enumerate(rots):
                o, sp = orc.get_cur_templ(lib, P[ip], rot)
                k = 'curtempl/%s/p%d_r%d/' % (name, ip, ir)
                np.testing.assert_allclose(sp, cases[k + 'spec'], rtol=3e-7)
                assert abs(o - cases[k + 'outside']) < 1e-12


def test_bases(cases):
    lam = cases['basis/lam']
    for key, n, rbf in (('rbf10', 10, True), ('rbf15', 15, True),
                        ('rbf2', 2, True), ('cheb7', 7, False)):
        np.testing.assert_allclose(orc.get_poly_basis(lam, n, rbf),
                                   cases['basis/' + key], rtol=1e-13,
                                   atol=1e-15)


@pytest.mark.parametrize('tag', TAGS)
@pytest.mark.parametrize('use_c', [False, True])
def test_get_chisq(cases, gold_libs, gold_config, tag, use_c):
    sds = gold_specdata(cases, tag, orc.SpecData)
    for i in range(7):
        k = '%s/chisq/t%d/' % (tag, i)
        vs = float(cases[k + 'vsini'])
        rot = None if np.isnan(vs) else (vs, )
        opt = dict(npoly=int(cases[k + 'npoly']),
                   rbf_continuum=bool(cases[k + 'rbf']))
        val = orc.get_chisq(sds, float(cases[k + 'vel']), cases[k + 'param'],
                            rot, options=opt, config=gold_config,
                            libs=gold_libs, use_c=use_c)
        ref = float(cases[k + 'value'])
        assert abs(val - ref) <= 1e-7 * abs(ref), (i, val, ref)
        if i < 3:
            full = orc.get_chisq(sds, float(cases[k + 'vel']),
                                 cases[k + 'param'], rot, options=opt,
                                 config=gold_config, libs=gold_libs,
                                 full_output=True, use_c=use_c)
            assert abs(full['chisq'] - cases[k + 'full_chisq']) <= 1e-7 * abs(ref)
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
    val = orc.get_chisq(sds, float(cases[tag + '/vel']), cases[tag + '/truth'],
                        None, options=dict(npoly=10), config=gold_config,
                        libs=gold_libs, espec_systematic=0.05, use_c=use_c)
    assert abs(val - cases[tag + '/chisq/sys005']) <= 1e-7 * abs(val)


@pytest.mark.parametrize('tag', TAGS)
def test_find_best(cases, gold_libs, gold_config, tag):
    sds = gold_specdata(cases, tag, orc.SpecData)
    vg = cases['vel_grid']
    for g in ('g1', 'g3'):
        k = '%s/%s/' % (tag, g)
        vs = float(cases[k + 'vsini'])
        rot = None if np.isnan(vs) else (vs, )
        pl = [tuple(_) for _ in cases[k + 'params_list']]
        r = orc.find_best(sds, vg, pl, rot, options=dict(npoly=10),
                          config=gold_config, libs=gold_libs)
        refgrid = cases[k + 'chisq_grid']
        assert rel(r['chisq_grid'], refgrid) < 1e-7
        assert abs(r['best_vel'] - cases[k + 'best_vel']) < 1e-4
        assert abs(r['vel_err'] - cases[k + 'vel_err']) < 1e-5
        assert abs(r['kurtosis'] - cases[k + 'kurtosis']) < 1e-5
        assert abs(r['skewness'] - cases[k + 'skewness']) < 1e-5
        np.testing.assert_allclose(r['best_param'], cases[k + 'best_param'])
        np.testing.assert_allclose(r['probs'], cases[k + 'probs'], rtol=1e-5,
                                   atol=1e-12)
        # moments on the reference's own grid: exact restatement
        s = orc.grid_summary(vg, refgrid)
        assert abs(s['best_vel'] - cases[k + 'best_vel']) < 1e-10
        assert abs(s['vel_err'] - cases[k + 'vel_err']) < 1e-10
        if g == 'g1':
            fast = orc.chisq_grid_fast(sds, vg, pl[0], rot, dict(npoly=10),
                                       gold_config, gold_libs)
            assert rel(fast, refgrid[:, 0]) < 1e-7


@pytest.mark.parametrize('tag', TAGS)
def test_chisq_continuum(cases, tag):
    sds = gold_specdata(cases, tag, orc.SpecData)
    r = orc.get_chisq_continuum(sds, options=dict(npoly=10))
    np.testing.assert_allclose(r['chisq_array'], cases[tag + '/cont/chisq_array'],
                               rtol=1e-8)
    np.testing.assert_allclose(r['redchisq_array'],
                               cases[tag + '/cont/redchisq_array'], rtol=1e-8)


@pytest.mark.parametrize('tag', TAGS)
def test_ccf(cases, gold_libs, gold_config, tag):
    sds = gold_specdata(cases, tag, orc.SpecData)
    for sd in sds:
        k = '%s/ccf/%s/' % (tag, sd.name)
        ps, pi, info = orc.preprocess_data(sd.lam, sd.spec, sd.espec,
                                           gold_libs[sd.name].ccf,
                                           badmask=sd.badmask, details=True)
        np.testing.assert_allclose(info['filled'], cases[k + 'cont_spec0'],
                                   rtol=1e-13)
        np.testing.assert_allclose(info['cespec'], cases[k + 'cont_espec0'],
                                   rtol=1e-13)
        np.testing.assert_allclose(info['p0'], cases[k + 'cont_p0'], rtol=1e-12)
        # scipy.optimize.least_squares is third-party arithmetic at the
        # boundary (scipy 1.7 in the capture interpreter, 1.15 here): the TRF
        # iterates agree only to its own 1e-8 tolerances.
        np.testing.assert_allclose(info['x'], cases[k + 'cont_x'], rtol=0,
                                   atol=2e-6)
        np.testing.assert_allclose(ps, cases[k + 'proc_spec'], rtol=1e-5,
                                   atol=1e-7)
        np.testing.assert_allclose(pi, cases[k + 'proc_ivar'], rtol=1e-5)
    r = orc.ccf_fit(sds, gold_config, gold_libs)
    np.testing.assert_allclose(r['best_par'], cases[tag + '/ccf/best_par'])
    assert abs(r['best_vel'] - cases[tag + '/ccf/best_vel']) < 1e-3
    np.testing.assert_allclose(r['best_ccf'], cases[tag + '/ccf/best_ccf'],
                               rtol=1e-6)
    np.testing.assert_allclose(r['vel_grid'], cases[tag + '/ccf/vel_grid'])
    bv = float(cases[tag + '/ccf/best_vsini'])
    assert r['best_vsini'] == bv
    for sd in sds:
        np.testing.assert_allclose(
            r['best_model'][sd.name],
            cases['%s/ccf/%s/best_model' % (tag, sd.name)])


def test_ccf_lag_tables_are_integer_exact(gold_libs):
    cc = gold_libs['gold_b'].ccf
    step, ind, sub = orc.ccf_lag_tables(cc['logl0'], cc['logl1'], cc['npoints'],
                                        1000)
    assert ind.dtype.kind == 'i' and len(ind) % 2 == 1
    assert np.all(np.diff(sub) > 0)
    # centre lag is lag 0
    assert ind[len(ind) // 2] == 0


@pytest.mark.parametrize('tag', TAGS)
def test_refine(cases, gold_libs, gold_config, tag):
    sds = gold_specdata(cases, tag, orc.SpecData)
    bv, be, sk, ku, grids = orc.find_best_vel_iterate(
        float(cases[tag + '/refine/start_vel']), gold_config, sds,
        tuple(cases[tag + '/truth']), None, dict(npoly=10), gold_libs)
    assert len(grids) == int(cases[tag + '/refine/ngrids'])
    for i, g in enumerate(grids):
        np.testing.assert_allclose(g, cases['%s/refine/grid_%d' % (tag, i)],
                                   rtol=0, atol=1e-5)
    assert abs(bv - cases[tag + '/refine/best_vel']) < 1e-4
    assert abs(be - cases[tag + '/refine/vel_err']) < 1e-5
    assert abs(sk - cases[tag + '/refine/skewness']) < 1e-4
    assert abs(ku - cases[tag + '/refine/kurtosis']) < 1e-4


def test_firstguess(cases, gold_libs, gold_config):
    sds = gold_specdata(cases, 'c0', orc.SpecData)
    pg = {'logg': [1, 3], 'teff': [4000, 5000, 7000], 'feh': [-2, -1],
          'alpha': [0]}
    fg = orc.firstguess(sds, dict(npoly=10), gold_config, gold_libs,
                        vsinigrid=(None, 100), paramsgrid=pg)
    keys = [str(_) for _ in cases['c0/firstguess/keys']]
    assert sorted(fg.keys()) == keys
    np.testing.assert_allclose([float(fg[k]) for k in keys],
                               cases['c0/firstguess/vals'])


def test_nn_forward_vs_reference_golden():
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'nn_case.npz'))
    W = [(d['W%d' % i], d['b%d' % i]) for i in range(5)]
    out = orc.nn_forward(W, d['params'], d['M'], d['S'])
    # float32 network: summation order of the matmuls differs (BLAS vs torch)
    np.testing.assert_allclose(out, d['out'], rtol=2e-6)


# --------------------------------------------------------------------------
# vel_fit.process
# --------------------------------------------------------------------------
@pytest.mark.parametrize('t', ['p0', 'p1', 'p2', 'p3'])
def test_process_oracle_vs_reference(cases, gold_libs, gold_config, t):
    """the oracle's process (scipy Nelder-Mead on the oracle's get_chisq) lands
    where the reference's own run did; param_err uses the same stand-in Hessian
    rule as the golden harness (numdifftools absent: unpinned)"""
    import os
    from conftest import GOLD
    g = np.load(os.path.join(GOLD, 'process_cases.npz'))
    sds = gold_specdata(cases, str(g[t + '/case']), orc.SpecData)
    pd0 = dict(zip([str(_) for _ in g[t + '/start_keys']],
                   [float(_) for _ in g[t + '/start_vals']]))
    fix = [str(_) for _ in g[t + '/fix']]
    pri = None
    if t + '/prior_keys' in g:
        pri = {str(k): tuple(v) for k, v in zip(g[t + '/prior_keys'],
                                                g[t + '/prior_vals'])}
    r = orc.process(sds, pd0, fix, dict(npoly=10), gold_config, gold_libs,
                    priors=pri)
    assert abs(r['nm_nit'][-1] - int(g[t + '/nm_nit'][-1])) <= 5
    assert abs(r['vel'] - g[t + '/vel']) < 1e-4
    assert abs(r['chisq'] - g[t + '/chisq']) < 1e-6
    assert abs(r['vel_err'] / g[t + '/vel_err'] - 1) < 1e-5
    err = g[t + '/param_err']
    ok = np.isfinite(err) & (err > 0)
    got = np.array(list(r['param'].values()))
    assert np.all(np.abs(got - g[t + '/param'])[ok] < 1e-3 * err[ok] + 1e-9)
    gerr = np.array(list(r['param_err'].values()))
    np.testing.assert_allclose(gerr[ok], err[ok], rtol=1e-3)
    assert r['bad_hessian'] == bool(g[t + '/bad_hessian'])
    assert r['minimize_success'] == bool(g[t + '/minimize_success'])


# --------------------------------------------------------------------------
# A9: resolution matrices
# --------------------------------------------------------------------------
def _dia(g, key, n):
    import scipy.sparse
    return scipy.sparse.dia_matrix((g[key + '/data'], g[key + '/offsets']),
                                   shape=(n, n))


@pytest.mark.parametrize('tag', ['c1', 'c2'])
def test_resolution_matrix_oracle(cases, gold_libs, gold_config, tag):
    import os
    from conftest import GOLD
    g = np.load(os.path.join(GOLD, 'resol_cases.npz'))
    names = [str(_) for _ in cases[tag + '/names']]
    sds = gold_specdata(cases, tag, orc.SpecData)
    opt = dict(npoly=10)
    # the oracle's construct_resol_mat against the reference's matrices
    rp = {}
    for sd in sds:
        n = len(sd.lam)
        ref = _dia(g, '%s/rp/%s' % (tag, sd.name), n)
        mine = orc.construct_resol_mat(sd.lam, resol=2500.)
        np.testing.assert_allclose(mine.toarray(), ref.toarray(), rtol=1e-13,
                                   atol=1e-300)
        rp[sd.name] = ref
    for i in range(3):
        vs = float(g['%s/rp/t%d/vsini' % (tag, i)])
        rot = None if np.isnan(vs) else (vs, )
        val = orc.get_chisq(sds, float(g['%s/rp/t%d/vel' % (tag, i)]),
                            tuple(g['%s/rp/t%d/param' % (tag, i)]), rot,
                            options=opt, config=gold_config, libs=gold_libs,
                            resol_params=rp)
        want = float(g['%s/rp/t%d/value' % (tag, i)])
        assert abs(val - want) < 1e-8 * max(abs(want), 1e3)
    fb = orc.find_best(sds, g['vel_grid'], [tuple(cases[tag + '/truth'])],
                       options=opt, config=gold_config, libs=gold_libs,
                       resol_params=rp)
    np.testing.assert_allclose(fb['chisq_grid'][:, 0], g[tag + '/rp/grid'],
                               rtol=1e-8)
    np.testing.assert_allclose([fb['best_vel'], fb['vel_err'], fb['best_chi']],
                               g[tag + '/rp/find_best'], rtol=1e-7, atol=1e-9)
    # per-spectrum matrices + continuum
    sds2 = [orc.SpecData(sd.name, sd.lam, sd.spec, sd.espec, badmask=sd.badmask,
                         resolution=_dia(g, '%s/own/%s' % (tag, sd.name),
                                         len(sd.lam))) for sd in sds]
    for i in range(3):
        vs = float(g['%s/rp/t%d/vsini' % (tag, i)])
        rot = None if np.isnan(vs) else (vs, )
        val = orc.get_chisq(sds2, float(g['%s/rp/t%d/vel' % (tag, i)]),
                            tuple(g['%s/rp/t%d/param' % (tag, i)]), rot,
                            options=opt, config=gold_config, libs=gold_libs)
        want = float(g['%s/own/t%d/value' % (tag, i)])
        assert abs(val - want) < 1e-8 * max(abs(want), 1e3)
    c = orc.get_chisq_continuum(sds2, options=opt)
    np.testing.assert_allclose(c['chisq_array'], g[tag + '/own/cont/chisq_array'],
                               rtol=1e-9)


# --------------------------------------------------------------------------
# Delaunay (triangulation) evaluator: spec_inter.TriInterp
# --------------------------------------------------------------------------
def test_triangulation_oracle(cases):
    import os
    from conftest import GOLD, GOLD_CONFIG
    g = np.load(os.path.join(GOLD, 'tri_cases.npz'))
    libs = {n: orc.TriLibrary(np.load(os.path.join(GOLD, 'lib_tri_%s.npz' % n)))
            for n in ('gold_b', 'gold_r')}
    P = g['params']
    for n, lib in libs.items():
        for i, p in enumerate(P):
            with np.errstate(all='ignore'):
                spec, info = lib.eval(p, details=True)
                o = lib.outside_flag(p)
            ref = g[n + '/eval'][i]
            if int(g[n + '/simplex'][i]) < 0:
                assert info['simplex'] == -1 and np.isnan(o)
                continue
            np.testing.assert_allclose(spec, ref, rtol=1e-13)
            assert abs(o - g[n + '/outside'][i]) < 1e-12
    sds = gold_specdata(cases, 'c1', orc.SpecData)
    for i in range(4):
        vs = float(g['c1/t%d/vsini' % i])
        with np.errstate(all='ignore'):
            val = orc.get_chisq(sds, float(g['c1/t%d/vel' % i]),
                                tuple(g['c1/t%d/param' % i]),
                                None if np.isnan(vs) else (vs, ),
                                options=dict(npoly=10), config=GOLD_CONFIG,
                                libs=libs)
        want = float(g['c1/t%d/value' % i])
        assert abs(val - want) < 1e-8 * max(abs(want), 1e3)


# --------------------------------------------------------------------------
# the non-continuum-normalised CCF set (config['ccf_continuum_normalize']=False)
# --------------------------------------------------------------------------
@pytest.fixture(scope='module')
def nocont():
    return dict(np.load(os.path.join(GOLD, 'nocont_cases.npz')))


@pytest.mark.parametrize('tag', ['c0', 'c1', 'c2', 'c3'])
def test_ccf_nocontinuum_oracle_vs_reference(cases, nocont, gold_libs,
                                             gold_config, tag):
    """fitter_ccf.py:40-47, 204-207 and the no-continuum branches of
    preprocess_data (make_ccf.py:370-376) against the reference's own run with
    an `rvs_make_ccf --nocontinuum` template set"""
    sds = gold_specdata(cases, tag, orc.SpecData)
    cfg = dict(gold_config, ccf_continuum_normalize=False)
    for sd in sds:
        cc = gold_libs[sd.name].ccf_set(cfg)
        assert not cc['continuum']
        ps, pi = orc.preprocess_data(sd.lam, sd.spec, sd.espec, cc,
                                     badmask=sd.badmask)
        k = '%s/%s/' % (tag, sd.name)
        np.testing.assert_allclose(ps, nocont[k + 'proc_spec'], rtol=1e-11,
                                   atol=1e-11)   # (numpy 1.26 there, 2.2 here)
        np.testing.assert_allclose(pi, nocont[k + 'proc_ivar'], rtol=1e-11,
                                   atol=0)
    o = orc.ccf_fit(sds, cfg, gold_libs, details=True)
    assert o['best_id'] == int(nocont[tag + '/best_id'])
    np.testing.assert_allclose(o['all_chisqs'].min(axis=1),
                               nocont[tag + '/template_min'], rtol=1e-9)
    np.testing.assert_allclose(o['best_ccf'], nocont[tag + '/best_ccf'],
                               rtol=1e-9)
    assert abs(o['best_vel'] - float(nocont[tag + '/best_vel'])) < 1e-6
    np.testing.assert_allclose(o['best_par'], nocont[tag + '/best_par'])
    for sd in sds:
        np.testing.assert_array_equal(o['best_model'][sd.name],
                                      nocont['%s/%s/best_model' % (tag, sd.name)])
    # the default (key missing or None) is the continuum-normalised set
    assert gold_libs[sds[0].name].ccf_set(gold_config)['continuum']
    assert gold_libs[sds[0].name].ccf_set(
        dict(gold_config, ccf_continuum_normalize=None))['continuum']


@pytest.mark.parametrize('tag', ['c0', 'c1'])
@pytest.mark.parametrize('npoly', [17, 24, 32])
def test_npoly_above_16_vs_reference(gold_libs, gold_config, tag, npoly):
    """continuum bases of more than 16 functions (monomials + Gaussians at 17 and 32,
    Chebyshev at 24): the oracle's get_chisq / get_chisq_continuum against the
    reference's own values (npoly_wide_cases.npz, make_golden_npoly_wide.py)"""
    g = np.load(os.path.join(GOLD, 'npoly_wide_cases.npz'))
    cases = np.load(os.path.join(GOLD, 'cases.npz'))
    sds = gold_specdata(cases, tag, orc.SpecData)
    k0 = '%s/p%d/' % (tag, npoly)
    opt = dict(npoly=npoly, rbf_continuum=bool(g[k0 + 'rbf']))
    c = orc.get_chisq_continuum(sds, options=opt)
    np.testing.assert_allclose(c['chisq_array'], g[k0 + 'cont/chisq_array'], rtol=1e-8)
    npix = sum(len(_.lam) for _ in sds)
    for ip in range(2):
        k = k0 + 't%d/' % ip
        vs = float(g[k + 'vsini'])
        rot = None if np.isnan(vs) else (vs, )
        full = orc.get_chisq(sds, float(g[k + 'vel']), tuple(g[k + 'param']),
                             rot_params=rot, options=opt, config=gold_config,
                             libs=gold_libs, full_output=True)
        # (-2 log L passes near zero: scaled by the pixel count)
        assert abs(full['chisq'] - float(g[k + 'value'])) <= 1e-7 * max(
            abs(float(g[k + 'value'])), npix)
        np.testing.assert_allclose(full['chisq_array'], g[k + 'chisq_array'],
                                   rtol=1e-7)
        np.testing.assert_array_equal(full['npix_array'], g[k + 'npix_array'])


@pytest.mark.parametrize('tag,npoly', [('c0', 17), ('c1', 24)])
def test_find_best_above_16_vs_reference(gold_libs, gold_config, tag, npoly):
    """the oracle's find_best at 17 / 24 continuum functions against the reference's
    own (npoly_wide_grid_cases.npz, make_golden_npoly_wide.py grid_callers)"""
    g = np.load(os.path.join(GOLD, 'npoly_wide_grid_cases.npz'))
    cases = np.load(os.path.join(GOLD, 'cases.npz'))
    sds = gold_specdata(cases, tag, orc.SpecData)
    k0 = '%s/p%d/' % (tag, npoly)
    opt = dict(npoly=npoly, rbf_continuum=bool(g[k0 + 'rbf']))
    vg = g[k0 + 'vel_grid'][20:41]      # (a third of the grid: CPU seconds)
    pl = [tuple(_) for _ in g[k0 + 'params']][:2]
    npix = sum(len(_.lam) for _ in sds)
    k = k0 + 'rot/'
    chi = np.array([[float(orc.get_chisq(sds, v, par, rot_params=(20., ), options=opt,
                                         config=gold_config, libs=gold_libs))
                     for par in pl] for v in vg])
    want = g[k + 'chisq'][20:41, :2]
    assert np.abs(chi - want).max() <= 1e-7 * max(np.abs(want).max(), npix)
