#!/opt/conda/bin/python3.9
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Run in the build container only (the reference never travels to the GPU box):

    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden.py

Recipe: a scratch copy of
/root/reference/py/rvspecfit lives in /tmp/oracle with a `_version.py` stub and
the cffi spline module compiled from the reference's own spliner.c; numba is
masked (broken in this image -> the reference takes its SVD branch, which is
mathematically identical to the numba-Cholesky production branch,
spec_fit.py:207-229) and numdifftools is stubbed (only used for the Hessian).

What is written (data only: inputs + reference outputs):
  lib_<setup>.npz   the reference-format artefacts of a small synthetic grid
                    produced by the REFERENCE prep pipeline from FITS files
                    written by the build's own generator (rvspecfit_amd.synth)
  cases.npz         seeded inputs and the reference's outputs for every row of
                    the hot path
"""
import os
import sys
import types
import shutil
import pickle

os.environ['OMP_NUM_THREADS'] = '1'
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(HERE + '/../..')
sys.path.insert(0, '/tmp/oracle')
sys.path.insert(0, REPO)
sys.modules['numba'] = None
sys.modules['numdifftools'] = types.ModuleType('numdifftools')

import numpy as np  # noqa: E402
import scipy.optimize  # noqa: E402
from rvspecfit import (spec_fit, spec_inter, fitter_ccf, make_ccf, vel_fit,  # noqa
                       utils, spliner, read_grid, make_interpol, make_nd,
                       serializer)
from rvspecfit_amd import synth  # noqa: E402

WORK = '/tmp/golden_work'
TEMPL = WORK + '/templ/'

SETUPS = {
    'gold_b': dict(obs=(4400., 4720.1, 0.8), templ=(4380., 4740., 0.4),
                   ccf=(4400., 4720., 0.4)),
    'gold_r': dict(obs=(4700., 4940.1, 0.8), templ=(4680., 4960., 0.4),
                   ccf=(4700., 4940., 0.4)),
}
GRID_KW = dict(nteff=4, nlogg=4, nfeh=4, nalpha=4, teff_range=(3500., 7500.),
               logg_range=(1., 4.), feh_range=(-2., 0.), alpha_range=(0., 0.4))
# NB: the reference serializer needs equal-length uvecs (ragged lists fail
# with numpy>=1.24), hence the 4^4 grid.
HOLES = (37, 207)


def build_reference_artefacts():
    if os.path.exists(WORK):
        shutil.rmtree(WORK)
    os.makedirs(TEMPL)
    pref = WORK + '/hr/'
    os.makedirs(pref)
    lam_hr = np.linspace(4300, 5040, 37001)
    synth.write_fits_grid(pref, 'wave.fits', grid_kw=GRID_KW, holes=HOLES,
                          lam_hr=lam_hr)
    db = WORK + '/files.db'
    read_grid.main(['--prefix', pref, '--templdb', db])
    for name, S in SETUPS.items():
        l0, l1, st = S['templ']
        make_interpol.main([
            '--templdb', db, '--wavefile', pref + 'wave.fits', '--templprefix',
            pref, '--resol', '2000', '--lambda0',
            str(l0), '--lambda1',
            str(l1), '--step',
            str(st), '--setup', name, '--oprefix', TEMPL, '--nthreads', '1'
        ])
        make_nd.main(['--setup', name, '--prefix', TEMPL, '--regulargrid'])
        c0, c1, cs = S['ccf']
        make_ccf.main([
            '--setup', name, '--prefix', TEMPL, '--lambda0',
            str(c0), '--lambda1',
            str(c1), '--step',
            str(cs), '--every', '20', '--vsinis', '0,100', '--oprefix', TEMPL,
            '--nthreads', '1'
        ])
    with open(WORK + '/config.yaml', 'w') as fp:
        fp.write("template_lib: '%s'\nmin_vel: -1000\nmax_vel: 1000\n"
                 "min_vel_step: 0.2\nvel_step0: 5\nmin_vsini: 0.1\n"
                 "max_vsini: 500\n" % TEMPL)


def export_library(name):
    fd = serializer.load_dict_from_hdf5(TEMPL + make_nd.INTERPOL_H5_NAME % name)
    dats = np.load(TEMPL + make_nd.INTERPOL_DAT_NAME % name)
    ci = serializer.load_dict_from_hdf5(TEMPL + make_ccf.get_ccf_info_name(name))
    cd = np.load(TEMPL + make_ccf.get_ccf_dat_name(name))
    cm = np.load(TEMPL + make_ccf.get_ccf_mod_name(name))
    cc = ci['ccfconf']
    vs = np.array([np.nan if _ is None else _ for _ in ci['vsinis']],
                  dtype=float)
    out = dict(
        lam=fd['lam'], dats=dats, vec=fd['vec'], idgrid=fd['idgrid'],
        log_step=np.array(bool(fd['log_step'])),
        log_ids=np.array(fd['mapper_args'][0], dtype=np.int64),
        parnames=np.array(list(fd['parnames'])),
        ccf_fft=cd['fft'], ccf_fft2=cd['fft2'], ccf_mod=cm,
        ccf_params=np.asarray(ci['params'], dtype=float), ccf_vsinis=vs,
        ccf_parnames=np.array(list(ci['parnames'])),
        ccf_logl0=np.array(cc['logl0']), ccf_logl1=np.array(cc['logl1']),
        ccf_npoints=np.array(cc['npoints']),
        ccf_continuum=np.array(bool(cc['continuum'])),
        ccf_splinestep=np.array(cc['splinestep']),
        ccf_maxcontpts=np.array(cc['maxcontpts']))
    for i, u in enumerate(fd['uvecs']):
        out['uvec%d' % i] = np.asarray(u)
    np.savez_compressed(HERE + '/lib_%s.npz' % name, **out)


class Rec:
    """Collect named arrays; nested keys are joined with '/'."""

    def __init__(self):
        self.d = {}

    def put(self, key, val):
        assert key not in self.d, key
        self.d[key] = np.asarray(val)


def obs_lam(name):
    a, b, c = SETUPS[name]['obs']
    return np.arange(a, b, c)


def make_specdata(names, truth, vel, snr, seed, mask_frac=0.0, slope=0.0):
    rng = np.random.RandomState(seed)
    sds = []
    raw = []
    for name in names:
        lam = obs_lam(name)
        spec, espec = synth.fake_observation(lam, *truth, vel, snr, rng,
                                             wresol=4700. / 2000 / 2.35,
                                             slope=slope)
        # divide by a smooth number so fluxes are O(1) like real data
        badmask = np.zeros(len(lam), dtype=bool)
        if mask_frac > 0:
            badmask = rng.uniform(size=len(lam)) < mask_frac
            espec = espec.copy()
            espec[badmask] *= 1e4
        sds.append(spec_fit.SpecData(name, lam, spec, espec, badmask=badmask))
        raw.append((lam, spec, espec, badmask))
    return sds, raw


def main():
    if '--reuse' not in sys.argv:
        build_reference_artefacts()
    for name in SETUPS:
        export_library(name)
    config = utils.read_config(WORK + '/config.yaml')
    R = Rec()

    # ---------------- A7: spline (reference C through cffi) -------------
    rng = np.random.RandomState(11)
    xs = np.exp(np.linspace(np.log(4000.), np.log(4400.), 300))
    ys = np.sin(xs / 7.) + 0.1 * rng.standard_normal(len(xs)) + 2
    S = spliner.Spline(xs, ys, log_step=True)
    ex = np.sort(rng.uniform(xs[0], xs[-1] * (1 - 1e-12), size=1000))
    ex[:50] = xs[3:53]  # exactly at knots
    ex = np.sort(ex)
    R.put('spline/log/xs', xs)
    R.put('spline/log/ys', ys)
    for k in 'ABCDh':
        R.put('spline/log/' + k, getattr(S, k))
    R.put('spline/log/evalx', ex)
    R.put('spline/log/ret', S(ex))
    xs = np.linspace(1000., 2000., 257)
    ys = 1e-5 * xs**2 + rng.standard_normal(len(xs))
    S = spliner.Spline(xs, ys, log_step=False)
    ex = np.sort(rng.uniform(1000, 1999.999, size=500))
    R.put('spline/lin/xs', xs)
    R.put('spline/lin/ys', ys)
    for k in 'ABCDh':
        R.put('spline/lin/' + k, getattr(S, k))
    R.put('spline/lin/evalx', ex)
    R.put('spline/lin/ret', S(ex))

    # ---------------- A3/A5: interpolator ------------------------------
    plist = [
        (5000., 2., -1., 0.2),      # in grid
        (4500., 1., -2., 0.0),      # exactly on nodes
        (7499., 3.99, -0.01, 0.39),  # near upper corner
        (6123., 2.5, -0.4, 0.1),
        (3400., 2., -1., 0.2),      # teff below grid -> nearest
        (8000., 5., 0.5, 0.6),      # far outside
        (5200., 2.2, -1.2, 0.45),   # alpha just outside
        (3700., 1.2, -1.9, 0.05),
        (5500., 3., -1., 0.2),      # possibly near a hole
        (4500., 3., -1., 0.),
        (5000., 2.7, -0.6, 0.3),
        (-100., 2., -1., 0.2),      # log10 of negative -> nan -> dats[0]
    ]
    # add the parameter cells touching the holes
    u, physvec = synth.regular_grid(**GRID_KW)
    for h in HOLES:
        p = physvec[:, h].copy()
        p[0] += 60.
        p[1] += 0.1
        p[2] += 0.07
        p[3] += 0.02
        plist.append(tuple(p))
    R.put('interp/params', np.array(plist))
    for name in SETUPS:
        it = spec_inter.getInterpolator(name, config)
        evs, outs, nearest = [], [], []
        for p in plist:
            with np.errstate(all='ignore'):
                evs.append(np.asarray(it.eval(p), dtype=np.float64))
                outs.append(float(it.outsideFlag(p)))
                mp = it.mapper.forward(p)
                nearest.append(
                    it.interper.get_nearest(mp) if np.isfinite(mp).all() else 0)
        R.put('interp/%s/eval' % name, np.array(evs))
        R.put('interp/%s/outside' % name, np.array(outs))
        R.put('interp/%s/nearest' % name, np.array(nearest))

    # ---------------- A6: vsini kernel + convolution --------------------
    Rs = [1e-3, 0.3, 0.9999, 1.0, 2.5, 7.123, 18.9, 36.5]
    R.put('vsini/R', np.array(Rs))
    for i, r in enumerate(Rs):
        R.put('vsini/kernel_%d' % i, spec_fit.compute_vsini_kernel(r))
    it = spec_inter.getInterpolator('gold_b', config)
    tspec = np.asarray(it.eval(plist[0]), dtype=np.float64)
    vs = [0., 1e-7, 5., 30., 100., 300., 500.]
    R.put('vsini/vsinis', np.array(vs))
    R.put('vsini/templ', tspec)
    for i, v in enumerate(vs):
        R.put('vsini/conv_%d' % i, spec_fit.convolve_vsini(it.lam, tspec, v))

    # getCurTempl with rotation
    rots = [None, (10.,), (300.,)]
    for name in SETUPS:
        for ip in (0, 3, 4):
            for ir, rot in enumerate(rots):
                o, lam_t, sp, tag, ls = spec_fit.getCurTempl(
                    name, tuple(plist[ip]), rot, config)
                R.put('curtempl/%s/p%d_r%d/spec' % (name, ip, ir), sp)
                R.put('curtempl/%s/p%d_r%d/outside' % (name, ip, ir), o)

    # ---------------- A2: bases -----------------------------------------
    lamb = obs_lam('gold_b')
    R.put('basis/lam', lamb)
    R.put('basis/rbf10', spec_fit.get_poly_basis(lamb, 10, rbf=True))
    R.put('basis/rbf15', spec_fit.get_poly_basis(lamb, 15, rbf=True))
    R.put('basis/rbf2', spec_fit.get_poly_basis(lamb, 2, rbf=True))
    R.put('basis/cheb7', spec_fit.get_poly_basis(lamb, 7, rbf=False))

    # ---------------- A10/A11/A12/A13/A14/A15/A16: spectra --------------
    cases = [
        dict(tag='c0', names=['gold_b'], truth=(5000., 2., -1., 0.2), vel=37.3,
             snr=100., seed=101, mask=0.0, slope=0.0),
        dict(tag='c1', names=['gold_b', 'gold_r'], truth=(6123., 2.5, -0.4, 0.1),
             vel=-212.7, snr=30., seed=102, mask=0.05, slope=0.3),
        dict(tag='c2', names=['gold_r'], truth=(4200., 3.5, -1.7, 0.3),
             vel=402.1, snr=1000., seed=103, mask=0.0, slope=-0.2),
        dict(tag='c3', names=['gold_b', 'gold_r'], truth=(7000., 1.5, -0.2, 0.0),
             vel=5.5, snr=10., seed=104, mask=0.02, slope=0.0),
    ]
    R.put('cases/tags', np.array([c['tag'] for c in cases]))
    vel_grid = np.arange(config['min_vel'], config['max_vel'],
                         config['vel_step0'])
    R.put('vel_grid', vel_grid)
    for c in cases:
        t = c['tag']
        sds, raw = make_specdata(c['names'], c['truth'], c['vel'], c['snr'],
                                 c['seed'], c['mask'], c['slope'])
        R.put(t + '/names', np.array(c['names']))
        R.put(t + '/truth', np.array(c['truth']))
        R.put(t + '/vel', c['vel'])
        for name, (lam, spec, espec, bm) in zip(c['names'], raw):
            R.put('%s/%s/lam' % (t, name), lam)
            R.put('%s/%s/spec' % (t, name), spec)
            R.put('%s/%s/espec' % (t, name), espec)
            R.put('%s/%s/badmask' % (t, name), bm)

        # get_chisq on a handful of (vel, param, rot, options)
        trials = [
            (c['vel'], c['truth'], None, dict(npoly=10)),
            (c['vel'] + 3.21, c['truth'], (30.,), dict(npoly=10)),
            (-871.3, plist[3], (300.,), dict(npoly=15)),
            (999.0, plist[0], None, dict(npoly=5)),
            (c['vel'], c['truth'], None, dict(npoly=7, rbf_continuum=False)),
            (12.5, plist[4], None, dict(npoly=10)),   # outside grid: penalty
            (12.5, plist[11], None, dict(npoly=10)),  # nan outside -> 1000*badchi
        ]
        for i, (v, p, rot, opt) in enumerate(trials):
            with np.errstate(all='ignore'):
                val = spec_fit.get_chisq(sds, v, p, rot, options=opt,
                                         config=config)
            R.put('%s/chisq/t%d/vel' % (t, i), v)
            R.put('%s/chisq/t%d/param' % (t, i), np.array(p))
            R.put('%s/chisq/t%d/vsini' % (t, i),
                  np.nan if rot is None else rot[0])
            R.put('%s/chisq/t%d/npoly' % (t, i), opt['npoly'])
            R.put('%s/chisq/t%d/rbf' % (t, i), opt.get('rbf_continuum', True))
            R.put('%s/chisq/t%d/value' % (t, i), val)
            if i < 3:
                full = spec_fit.get_chisq(sds, v, p, rot, options=opt,
                                          config=config, full_output=True)
                R.put('%s/chisq/t%d/full_chisq' % (t, i), full['chisq'])
                R.put('%s/chisq/t%d/chisq_array' % (t, i), full['chisq_array'])
                R.put('%s/chisq/t%d/red_chisq_array' % (t, i),
                      full['red_chisq_array'])
                R.put('%s/chisq/t%d/npix_array' % (t, i), full['npix_array'])
                for name, m, rm in zip(c['names'], full['models'],
                                       full['raw_models']):
                    R.put('%s/chisq/t%d/model_%s' % (t, i, name), m)
                    R.put('%s/chisq/t%d/raw_model_%s' % (t, i, name), rm)
        # espec_systematic
        val = spec_fit.get_chisq(sds, c['vel'], c['truth'], None,
                                 options=dict(npoly=10), config=config,
                                 espec_systematic=0.05)
        R.put(t + '/chisq/sys005', val)

        # chi^2 grid + find_best (1 template and 3 templates)
        opt = dict(npoly=10)
        for gtag, params_list, rot in (('g1', [c['truth']], None),
                                       ('g3', [plist[0], c['truth'], plist[3]],
                                        (30.,))):
            cache = spec_fit.LRUDict(100)
            grid = np.zeros((len(vel_grid), len(params_list)))
            for j, p in enumerate(params_list):
                for i, v in enumerate(vel_grid):
                    grid[i, j] = spec_fit.get_chisq(sds, v, p, rot, None,
                                                    options=opt, config=config,
                                                    cache=cache)
            fb = spec_fit.find_best(sds, vel_grid, params_list, rot_params=rot,
                                    options=opt, config=config)
            R.put('%s/%s/params_list' % (t, gtag), np.array(params_list))
            R.put('%s/%s/vsini' % (t, gtag), np.nan if rot is None else rot[0])
            R.put('%s/%s/chisq_grid' % (t, gtag), grid)
            for k in ('best_chi', 'best_vel', 'vel_err', 'kurtosis',
                      'skewness', 'probs'):
                R.put('%s/%s/%s' % (t, gtag, k), fb[k])
            R.put('%s/%s/best_param' % (t, gtag), np.array(fb['best_param']))

        # continuum-only chi^2
        cc = spec_fit.get_chisq_continuum(sds, options=opt)
        R.put(t + '/cont/chisq_array', cc['chisq_array'])
        R.put(t + '/cont/redchisq_array', cc['redchisq_array'])

        # CCF: record the continuum fit through wrappers (no reference edits)
        rec_cont = []
        orig_gc = make_ccf.get_continuum
        orig_ls = scipy.optimize.least_squares

        def wrap_gc(lam0, spec0, espec0, ccfconf=None):
            ret = orig_gc(lam0, spec0, espec0, ccfconf=ccfconf)
            rec_cont[-1].update(lam0=lam0.copy(), spec0=spec0.copy(),
                                espec0=espec0.copy(), cont=ret.copy())
            return ret

        def wrap_ls(fun, x0, **kw):
            ret = orig_ls(fun, x0, **kw)
            rec_cont.append(dict(p0=np.array(x0), x=ret['x'].copy(),
                                 cost=ret['cost'], nfev=ret['nfev']))
            return ret

        make_ccf.get_continuum = wrap_gc
        scipy.optimize.least_squares = wrap_ls
        try:
            res = fitter_ccf.fit(sds, config)
        finally:
            make_ccf.get_continuum = orig_gc
            scipy.optimize.least_squares = orig_ls
        for name, rc, sd in zip(c['names'], rec_cont, sds):
            for k, v in rc.items():
                R.put('%s/ccf/%s/cont_%s' % (t, name, k), v)
            ccfconf = fitter_ccf.get_ccf_info(name, config)[3]['ccfconf']
            ps, pi = make_ccf.preprocess_data(sd.lam, sd.spec, sd.espec,
                                              badmask=sd.badmask,
                                              ccfconf=ccfconf)
            R.put('%s/ccf/%s/proc_spec' % (t, name), ps)
            R.put('%s/ccf/%s/proc_ivar' % (t, name), pi)
            R.put('%s/ccf/%s/best_model' % (t, name), res['best_model'][name])
        R.put(t + '/ccf/best_vel', res['best_vel'])
        R.put(t + '/ccf/best_ccf', res['best_ccf'])
        R.put(t + '/ccf/best_vsini',
              np.nan if res['best_vsini'] is None else res['best_vsini'])
        R.put(t + '/ccf/vel_grid', res['vel_grid'])
        R.put(t + '/ccf/best_par',
              np.array([res['best_par'][k] for k in ('teff', 'logg', 'feh',
                                                     'alpha')]))

        # A16: velocity refinement, recording the grids it asks for
        grids = []
        orig_fb = spec_fit.find_best

        def wrap_fb(specdata, vg, *a, **kw):
            grids.append(np.array(vg))
            return orig_fb(specdata, vg, *a, **kw)

        spec_fit.find_best = wrap_fb
        try:
            bp = dict(params=tuple(c['truth']), rot_params=None)
            bv, be, sk, ku = vel_fit._find_best_vel_iterate(
                c['vel'] + 1.7, config['min_vel'], config['max_vel'],
                config['vel_step0'], specdata=sds, best_param=bp,
                config=config, options=opt,
                min_vel_step=config['min_vel_step'])
        finally:
            spec_fit.find_best = orig_fb
        R.put(t + '/refine/start_vel', c['vel'] + 1.7)
        R.put(t + '/refine/best_vel', bv)
        R.put(t + '/refine/vel_err', be)
        R.put(t + '/refine/skewness', sk)
        R.put(t + '/refine/kurtosis', ku)
        R.put(t + '/refine/ngrids', len(grids))
        for i, g in enumerate(grids):
            R.put('%s/refine/grid_%d' % (t, i), g)

    # firstguess on a reduced parameter grid (A16), case c0 only
    sds, raw = make_specdata(cases[0]['names'], cases[0]['truth'],
                             cases[0]['vel'], cases[0]['snr'], cases[0]['seed'])
    pg = {'logg': [1, 3], 'teff': [4000, 5000, 7000], 'feh': [-2, -1],
          'alpha': [0]}
    fg = vel_fit.firstguess(sds, options=dict(npoly=10), config=config,
                            vsinigrid=(None, 100), paramsgrid=pg)
    R.put('c0/firstguess/keys', np.array(sorted(fg.keys())))
    R.put('c0/firstguess/vals', np.array([float(fg[k]) for k in sorted(fg)]))

    np.savez_compressed(HERE + '/cases.npz', **R.d)
    print('wrote', len(R.d), 'arrays')
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(HERE + '/' + f))


if __name__ == '__main__':
    main()
