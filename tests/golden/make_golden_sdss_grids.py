#!/opt/conda/bin/python3.9
"""Golden vectors for BATCHES OF SPECTRA ON THEIR OWN WAVELENGTH GRIDS (build
container only; needs the scratch of tests/golden/setup_reference_scratch.sh and the
artefacts make_golden_sdss.py leaves in /tmp/golden_work_sdss):

    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_sdss_grids.py

The reference takes any `lam` per object (spec_fit.py:70-145) and fits SDSS-style
spectra one by one (tests/test_sdss.py).  Ten spectra are cut from the reference's
data fixture tests/data/spec-0266-51602-0031.fits on shifted / truncated pieces of
its log-lambda grid (two of them on the SAME piece), each with its own flux scale
and a seeded noise realisation, and every one goes through the reference ALONE:
fitter_ccf.fit, spec_fit.get_chisq at three (velocity, parameters, vsini) points,
find_best on a 201-point velocity grid, get_chisq_continuum, and -- for three of
them -- vel_fit.process.  Writes tests/golden/sdss_grid_cases.npz (inputs and the
reference's results; data only).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_sdss as mgs  # noqa: E402  (numpy shims, reference imports)
import make_golden as mg  # noqa: E402

import astropy.io.fits as pyfits  # noqa: E402
from rvspecfit import utils, spec_fit, vel_fit, fitter_ccf  # noqa: E402

# (first pixel, one past the last pixel) of the 3842-pixel grid; 3 and 7 share one
PIECES = [(0, 3842), (37, 3842), (120, 3700), (5, 3500), (400, 3842), (250, 3300),
          (0, 3000), (5, 3500), (800, 3842), (63, 3779)]
POINTS = [(30., (5000., 3., -1., 0.2), 19.), (-85., (5600., 4.2, -0.4, 0.1), None),
          (140., (4500., 2., -1.6, 0.3), 120.)]
PROCESS = (0, 2, 5)


def main():
    config = utils.read_config(mgs.WORK + '/config.yaml')
    dat = pyfits.getdata(mgs.SRC)
    err = dat['ivar']
    with np.errstate(all='ignore'):
        err = 1. / err**.5
    err[~np.isfinite(err)] = 1e30
    lam = 10**dat['loglam']
    flux = dat['flux']
    rng = np.random.RandomState(20261004)
    R = mg.Rec()
    R.put('pieces', np.array(PIECES))
    options = {'npoly': 10}
    vel_grid = np.linspace(-500, 500, 201)
    names = ('teff', 'logg', 'feh', 'alpha')
    for i, (a, b) in enumerate(PIECES):
        sc = 0.6 + 0.1 * i
        e = err[a:b] * sc
        noise = rng.normal(size=b - a) * np.where(e < 1e20, 0.3 * e, 0.0)
        sp = flux[a:b] * sc + noise
        sd = [spec_fit.SpecData('sdss1', lam[a:b], sp, e)]
        tag = 's%d' % i
        R.put(tag + '/lam', sd[0].lam)
        R.put(tag + '/spec', sd[0].spec)
        R.put(tag + '/espec', sd[0].espec)
        R.put(tag + '/badmask', sd[0].badmask)
        with np.errstate(all='ignore'):
            res = fitter_ccf.fit(sd, config)
            pd = dict(res['best_par'])
            R.put(tag + '/ccf/best_vel', res['best_vel'])
            R.put(tag + '/ccf/best_par', [pd[k] for k in names])
            R.put(tag + '/ccf/best_vsini', np.nan if res['best_vsini'] is None
                  else res['best_vsini'])
            R.put(tag + '/ccf/best_ccf', res['best_ccf'])
            for q, (v, par, vs) in enumerate(POINTS):
                ret = spec_fit.get_chisq(sd, v, par,
                                         rot_params=None if vs is None else (vs, ),
                                         config=config, options=options,
                                         full_output=True)
                R.put('%s/pt%d/chisq' % (tag, q), ret['chisq'])
                R.put('%s/pt%d/chisq_array' % (tag, q), ret['chisq_array'])
                R.put('%s/pt%d/model' % (tag, q), ret['models'][0])
            fb = spec_fit.find_best(sd, vel_grid, [list(POINTS[0][1]),
                                                   list(POINTS[1][1])],
                                    rot_params=None, resol_params=None,
                                    options=options, config=config)
            for k in ('best_vel', 'best_chi', 'vel_err', 'kurtosis', 'skewness'):
                R.put('%s/find_best/%s' % (tag, k), fb[k])
            R.put(tag + '/find_best/best_param', fb['best_param'])
            R.put(tag + '/continuum', spec_fit.get_chisq_continuum(
                sd, options=options)['chisq_array'])
            if i in PROCESS:
                if res['best_vsini'] is not None:
                    pd['vsini'] = res['best_vsini']
                mgs.put_fit(R, tag + '/process', vel_fit.process(
                    sd, dict(pd), fixParam=[], config=config, options=options))
        print(tag, 'done', flush=True)
    np.savez_compressed(HERE + '/sdss_grid_cases.npz', **R.d)
    print('wrote', len(R.d), 'arrays')


if __name__ == '__main__':
    main()
