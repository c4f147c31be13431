#!/opt/conda/bin/python3.9
"""Golden vectors for a BATCH OF SPECTRA ON THEIR OWN WAVELENGTH GRIDS THAT EACH
CARRY A RESOLUTION MATRIX (build container only; needs the scratch of
tests/golden/setup_reference_scratch.sh and the artefacts make_golden_sdss.py leaves
in /tmp/golden_work_sdss):

    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_sdss_resol.py

The reference accepts any SpecData.resolution (spec_fit.py:922-929; its own
tests/test_sdss.py fits the SDSS fixture through a resolution matrix).  The ten
spectra of sdss_grid_cases.npz (pieces of the fixture's log-lambda grid, two on the
same piece) each get spec_fit.construct_resol_mat(lam, R_i) of their own -- set A:
R = 1900 ... 2350, bands of 9 and 11 diagonals; set B (the first four): R = 1000 ...
1300, 17-21 diagonals -- and go through the reference ALONE: spec_fit.get_chisq at
three points (value, chisq_array, model), find_best on a 201-point velocity grid
(whole chi^2 grid for the first parameter set), get_chisq_continuum.  Writes
tests/golden/sdss_grid_resol_cases.npz (the resolution numbers and the reference's
results; the spectra themselves are those of sdss_grid_cases.npz).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_sdss as mgs  # noqa: E402  (numpy shims, reference imports)
import make_golden as mg  # noqa: E402

from rvspecfit import utils, spec_fit  # noqa: E402

POINTS = [(30., (5000., 3., -1., 0.2), 19.), (-85., (5600., 4.2, -0.4, 0.1), None),
          (140., (4500., 2., -1.6, 0.3), 120.)]
SETS = dict(A=[1900. + 50 * i for i in range(10)], B=[1000. + 100 * i for i in range(4)])


def main():
    config = utils.read_config(mgs.WORK + '/config.yaml')
    g = np.load(HERE + '/sdss_grid_cases.npz')
    R = mg.Rec()
    options = {'npoly': 10}
    vel_grid = np.linspace(-500, 500, 201)
    for sname, resols in SETS.items():
        R.put(sname + '/resol', np.array(resols))
        for i, rr in enumerate(resols):
            t0 = 's%d' % i
            lam = g[t0 + '/lam']
            rm = spec_fit.construct_resol_mat(lam, rr)
            sd = [spec_fit.SpecData('sdss1', lam, g[t0 + '/spec'], g[t0 + '/espec'],
                                    badmask=g[t0 + '/badmask'], resolution=rm)]
            tag = '%s/s%d' % (sname, i)
            R.put(tag + '/ndiag', len(rm.mat.offsets))
            with np.errstate(all='ignore'):
                for q, (v, par, vs) in enumerate(POINTS):
                    ret = spec_fit.get_chisq(
                        sd, v, par, rot_params=None if vs is None else (vs, ),
                        config=config, options=options, full_output=True)
                    R.put('%s/pt%d/chisq' % (tag, q), ret['chisq'])
                    R.put('%s/pt%d/chisq_array' % (tag, q), ret['chisq_array'])
                    R.put('%s/pt%d/model' % (tag, q), ret['models'][0])
                plist = [list(POINTS[0][1]), list(POINTS[1][1])]
                fb = spec_fit.find_best(sd, vel_grid, plist, rot_params=None,
                                        options=options, config=config)
                for k in ('best_vel', 'best_chi', 'vel_err', 'kurtosis', 'skewness'):
                    R.put('%s/find_best/%s' % (tag, k), fb[k])
                R.put(tag + '/find_best/best_param', fb['best_param'])
                R.put(tag + '/find_best/chisq0', np.array([float(spec_fit.get_chisq(
                    sd, v, plist[0], config=config, options=options))
                    for v in vel_grid]))
                R.put(tag + '/continuum', spec_fit.get_chisq_continuum(
                    sd, options=options)['chisq_array'])
            print(tag, 'ndiag', len(rm.mat.offsets), 'done', flush=True)
    R.put('vel_grid', vel_grid)
    np.savez_compressed(HERE + '/sdss_grid_resol_cases.npz', **R.d)
    print('wrote', len(R.d), 'arrays')


if __name__ == '__main__':
    main()
