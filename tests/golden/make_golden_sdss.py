#!/opt/conda/bin/python3.9
"""Golden vectors on the reference's own data fixture, tests/data/
spec-0266-51602-0031.fits (a real SDSS spectrum, 3842 log-spaced pixels), following
the reference's tests/test_fit.py and tests/test_sdss.py call by call (build
container only):

    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_sdss.py

test_fit.py needs a PHOENIX-derived template set ('tests/templ_data_sdss/') that is
not in the checkout; a synthetic 3^4 grid over the same wavelength range stands in
for it (prepared by the REFERENCE's own read_grid / make_interpol / make_nd /
make_ccf, as in make_golden.py).  What test_fit.py exercises and the other golden
sets do not: npoly = 15, the Chebyshev continuum (rbf_continuum=False), a prior,
real noise and a real continuum, a log-spaced observed grid of 3842 pixels.
numdifftools: the stand-in of make_golden_process.py (param_err unpinned).

Writes (data only):
  spec-0266-51602-0031.fits   copy of the reference's test data file (input)
  lib_sdss1.npz               reference-format artefacts of the synthetic grid
  sdss_cases.npz              the reference's results for every call of test_fit
"""
import os
import sys
import shutil

import numpy as np

for _n, _f in dict(asscalar=lambda a: a.item(), alen=len,
                   msort=lambda a: np.sort(a, axis=0), sometrue=np.any,
                   alltrue=np.all, product=np.prod, cumproduct=np.cumprod,
                   rank=np.ndim,
                   asfarray=lambda a, dtype=float: np.asarray(a, dtype=dtype)
                   ).items():
    if not hasattr(np, _n):
        setattr(np, _n, _f)
for _n, _t in dict(float=float, int=int, bool=bool, object=object,
                   complex=complex, str=str).items():
    if _n not in np.__dict__:
        setattr(np, _n, _t)

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_process as mgp  # noqa: E402,F401  (numdifftools stand-in)
import make_golden as mg  # noqa: E402

import astropy.io.fits as pyfits  # noqa: E402
from rvspecfit import (utils, read_grid, make_interpol, make_nd,  # noqa: E402
                       make_ccf, spec_fit, vel_fit, fitter_ccf)
from rvspecfit_amd import synth  # noqa: E402

WORK = '/tmp/golden_work_sdss'
TEMPL = WORK + '/templ/'
SRC = '/root/reference/tests/data/spec-0266-51602-0031.fits'
GRID_KW = dict(nteff=3, nlogg=3, nfeh=3, nalpha=3, teff_range=(4000., 7000.),
               logg_range=(1., 5.), feh_range=(-2., 0.), alpha_range=(0., 0.4))


def build_reference_artefacts():
    if os.path.exists(WORK):
        shutil.rmtree(WORK)
    os.makedirs(TEMPL)
    pref = WORK + '/hr/'
    os.makedirs(pref)
    lam_hr = np.arange(3700., 9400., 0.1)
    synth.write_fits_grid(pref, 'wave.fits', grid_kw=GRID_KW, lam_hr=lam_hr)
    db = WORK + '/files.db'
    read_grid.main(['--prefix', pref, '--templdb', db])
    make_interpol.main([
        '--templdb', db, '--wavefile', pref + 'wave.fits', '--templprefix', pref,
        '--resol', '2000', '--lambda0', '3750', '--lambda1', '9300', '--step',
        '1.0', '--setup', 'sdss1', '--oprefix', TEMPL, '--nthreads', '1'
    ])
    make_nd.main(['--setup', 'sdss1', '--prefix', TEMPL, '--regulargrid'])
    make_ccf.main([
        '--setup', 'sdss1', '--prefix', TEMPL, '--lambda0', '3800', '--lambda1',
        '9200', '--step', '2.0', '--every', '16', '--vsinis', '0,100',
        '--oprefix', TEMPL, '--nthreads', '1'
    ])
    with open(WORK + '/config.yaml', 'w') as fp:
        fp.write("template_lib: '%s'\nmin_vel: -1000\nmax_vel: 1000\n"
                 "min_vel_step: 0.2\nvel_step0: 5\nmin_vsini: 0.1\n"
                 "max_vsini: 500\n" % TEMPL)


def put_fit(R, tag, res):
    for k in ('vel', 'vel_err', 'vel_skewness', 'vel_kurtosis', 'chisq'):
        R.put('%s/%s' % (tag, k), res[k])
    R.put(tag + '/vsini', np.nan if res.get('vsini') is None else res['vsini'])
    names = ['teff', 'logg', 'feh', 'alpha']
    R.put(tag + '/param', [res['param'][k] for k in names])
    R.put(tag + '/param_err', [res['param_err'][k] for k in names])
    R.put(tag + '/chisq_array', res['chisq_array'])
    R.put(tag + '/npix_array', res['npix_array'])
    R.put(tag + '/yfit', res['yfit'][0])
    R.put(tag + '/bad_hessian', res['bad_hessian'])


def main():
    if '--reuse' not in sys.argv:
        build_reference_artefacts()
    old = mg.TEMPL
    mg.TEMPL = TEMPL
    try:
        mg.export_library('sdss1')
    finally:
        mg.TEMPL = old
    shutil.copy(SRC, HERE + '/spec-0266-51602-0031.fits')
    os.chmod(HERE + '/spec-0266-51602-0031.fits', 0o644)
    config = utils.read_config(WORK + '/config.yaml')
    R = mg.Rec()
    # ---- tests/test_fit.py:33-44
    dat = pyfits.getdata(SRC)
    err = dat['ivar']
    with np.errstate(all='ignore'):
        err = 1. / err**.5
    err[~np.isfinite(err)] = 1e30
    lam = 10**dat['loglam']
    specdata = [spec_fit.SpecData('sdss1', lam, dat['flux'], err)]
    R.put('data/lam', specdata[0].lam)
    R.put('data/spec', specdata[0].spec)
    R.put('data/espec', specdata[0].espec)
    options = {'npoly': 15}
    p0 = {'logg': 2, 'teff': 5000, 'feh': -1, 'alpha': 0.2, 'vsini': 19}
    with np.errstate(all='ignore'):
        # :60-66 fixed vsini
        put_fit(R, 'fixvsini', vel_fit.process(
            specdata, dict(p0), fixParam=['vsini'], config=config,
            options=options))
        # :77-83 vsini free
        put_fit(R, 'free', vel_fit.process(
            specdata, dict(p0), fixParam=[], config=config, options=options))
        # :88 first guess
        g = vel_fit.firstguess(specdata, config=config)
        R.put('firstguess/keys', np.array(sorted(g.keys())))
        R.put('firstguess/vals', np.array([float(g[k]) for k in sorted(g.keys())]))
        # :91-100 CCF start
        res = fitter_ccf.fit(specdata, config)
        pd = dict(res['best_par'])
        R.put('ccf/best_vel', res['best_vel'])
        R.put('ccf/best_par', [pd[k] for k in ('teff', 'logg', 'feh', 'alpha')])
        R.put('ccf/best_vsini', np.nan if res['best_vsini'] is None
              else res['best_vsini'])
        R.put('ccf/best_ccf', res['best_ccf'])
        if res['best_vsini'] is not None:
            pd['vsini'] = res['best_vsini']
        put_fit(R, 'ccfstart', vel_fit.process(
            specdata, dict(pd), fixParam=[], config=config, options=options))
        # :110-115 prior
        put_fit(R, 'prior', vel_fit.process(
            specdata, dict(pd), fixParam=[], config=config, options=options,
            priors={'teff': (9000, 50)}))
        # :117-122 Chebyshev continuum
        opt2 = dict(options)
        opt2['rbf_continuum'] = False
        put_fit(R, 'cheb', vel_fit.process(
            specdata, dict(pd), fixParam=[], config=config, options=opt2))
        # plus the objective itself at the start point, both bases
        for tag, op in (('rbf', options), ('chebb', opt2)):
            R.put('chisq0/' + tag, spec_fit.get_chisq(
                specdata, 30., tuple(float(p0[k]) for k in (
                    'teff', 'logg', 'feh', 'alpha')),
                rot_params=(19., ), config=config, options=op))
        R.put('continuum/chisq_array', spec_fit.get_chisq_continuum(
            specdata, options=options)['chisq_array'])
    # ================= tests/test_sdss.py, call by call =====================
    options = {'npoly': 10}
    params_list = [[4000, 3, -1, 0], [5000, 3, -1, 0], [6000, 2, -2, 0],
                   [5500, 5, 0, 0]]
    vel_grid = np.linspace(-600, 600, 1000)
    with np.errstate(all='ignore'):
        # :41-51 find_best on a 1000-point grid, 4 templates
        res = spec_fit.find_best(specdata, vel_grid, params_list,
                                 rot_params=None, resol_params=None,
                                 options=options, config=config)
        for k in ('best_vel', 'best_chi', 'vel_err', 'kurtosis', 'skewness'):
            R.put('t2/find_best/' + k, res[k])
        R.put('t2/find_best/best_param', res['best_param'])
        bestv, bestpar = res['best_vel'], res['best_param']
        # :53-58 first guess (npoly 10) + process
        param0 = vel_fit.firstguess(specdata, options=options, config=config)
        R.put('t2/firstguess/keys', np.array(sorted(param0.keys())))
        R.put('t2/firstguess/vals',
              np.array([float(param0[k]) for k in sorted(param0.keys())]))
        put_fit(R, 't2/process', vel_fit.process(
            specdata, dict(param0), resolParams=None, options=options,
            config=config))
        # :66-73 get_chisq with vsini 300
        ret = spec_fit.get_chisq(specdata, bestv, bestpar, rot_params=(300, ),
                                 resol_params=None, options=options,
                                 config=config, full_output=True)
        R.put('t2/rot300/chisq', ret['chisq'])
        R.put('t2/rot300/model', ret['models'][0])
        # :79-90 resolution matrix R = 50 through resol_params
        resol_mat = spec_fit.construct_resol_mat(specdata[0].lam, 50)
        R.put('t2/resol/ndiag', len(resol_mat.mat.offsets))
        rp = {'sdss1': resol_mat}
        ret = spec_fit.get_chisq(specdata, bestv, bestpar, None, resol_params=rp,
                                 options=options, config=config,
                                 full_output=True)
        R.put('t2/resol/chisq', ret['chisq'])
        R.put('t2/resol/chisq_array', ret['chisq_array'])
        R.put('t2/resol/model', ret['models'][0])
        # :95-100 process with resolParams
        put_fit(R, 't2/process_resol', vel_fit.process(
            specdata, dict(param0), resolParams=rp, options=options,
            config=config))
        # :101-117 the same matrix carried by the SpecData
        sd2 = [spec_fit.SpecData('sdss1', lam, dat['flux'], err,
                                 resolution=resol_mat)]
        ret = spec_fit.get_chisq(sd2, bestv, bestpar, None, options=options,
                                 config=config, full_output=True)
        R.put('t2/sdresol/chisq', ret['chisq'])
        R.put('t2/sdresol/model', ret['models'][0])
        # :121 continuum with the resolution matrix
        R.put('t2/sdresol/continuum', spec_fit.get_chisq_continuum(
            sd2, options=options)['chisq_array'])
    np.savez_compressed(HERE + '/sdss_cases.npz', **R.d)
    print('wrote', len(R.d), 'arrays')


if __name__ == '__main__':
    main()
