#!/opt/conda/bin/python3.9
"""The accuracy harness at the reference's DEFAULT optimiser configuration
(`utils.read_config`: second_minimizer = True, utils.py:26 -- what tests/accuracy.py
itself runs: it reads tests/yamls/test.yaml, which does not switch it off): the 200
spectra of the S/N 100 harness run through the REFERENCE's vel_fit.process one by one,
BFGS polish included (eight worker processes).  make_golden_accuracy.py is the same
with second_minimizer = False.

    bash tests/golden/setup_reference_scratch.sh
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_accuracy_bfgs.py

Writes accuracy_bfgs_cases.npz: all/n, all/vel, all/vel_err, all/chisq."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_accuracy as A  # noqa: E402  (imports the reference, numdiff stand-in)
import numpy as np  # noqa: E402

mg, utils, spec_fit, vel_fit, accuracy_suite = A.mg, A.utils, A.spec_fit, A.vel_fit, \
    A.accuracy_suite
_ST = {}


def _one(i):
    if not _ST:
        base = dict(utils.read_config(mg.WORK + '/config.yaml'))
        base.update(min_vel=-1500, max_vel=1500, second_minimizer=True)
        _ST['config'] = utils.freezeDict(base)
        _ST['lam'] = np.load(os.path.join(HERE, 'cases.npz'))['c0/gold_b/lam']
        _ST['data'] = accuracy_suite.make_spectra(_ST['lam'], A.N_ALL, A.SN)
    v0, truth, spec, espec = _ST['data']
    sd = [spec_fit.SpecData('gold_b', _ST['lam'], spec[i], espec[i])]
    with np.errstate(all='ignore'):
        r = vel_fit.process(sd, dict(logg=2.5, teff=5000., feh=-1., alpha=0.5),
                            config=_ST['config'], options=dict(npoly=10))
    return i, float(r['vel']), float(r['vel_err']), float(r['chisq'])


def main():
    import multiprocessing as mp
    with mp.get_context('fork').Pool(8) as pool:
        allr = sorted(pool.map(_one, range(A.N_ALL), chunksize=5))
    out = {'all/n': np.array(A.N_ALL), 'sn': np.array(A.SN),
           'all/vel': np.array([r[1] for r in allr]),
           'all/vel_err': np.array([r[2] for r in allr]),
           'all/chisq': np.array([r[3] for r in allr])}
    lam = np.load(os.path.join(HERE, 'cases.npz'))['c0/gold_b/lam']
    v0a = accuracy_suite.make_spectra(lam, A.N_ALL, A.SN)[0]
    dx = out['all/vel'] - v0a
    print('reference (second_minimizer on), %d spectra: median dx %.4f median err %.4f '
          'std dx %.4f std pull %.4f' % (A.N_ALL, np.median(dx),
                                         np.median(out['all/vel_err']), np.std(dx),
                                         np.std(dx / out['all/vel_err'])))
    np.savez_compressed(os.path.join(HERE, 'accuracy_bfgs_cases.npz'), **out)


if __name__ == '__main__':
    main()
