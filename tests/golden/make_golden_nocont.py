#!/opt/conda/bin/python3.9
"""Golden vectors for the NON-continuum-normalised CCF set
(config['ccf_continuum_normalize'] = False, fitter_ccf.py:40-47, 204-207;
rvs_make_ccf --nocontinuum, make_ccf.py:19-36, 94-97, 370-376, 530-561).

    bash tests/golden/setup_reference_scratch.sh
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_nocont.py

IMPORTS the reference (build container only).  The interpolation artefacts are
those of make_golden.py (rebuilt if /tmp/golden_work is gone); the reference's
own make_ccf writes the ccf_nocont_* files next to them.  Written:
  lib_nocont_<setup>.npz   the `ccfnc_*` arrays (what tools/convert_artefacts.py
                           adds to rvsgpu_<setup>.npz for the nocont set)
  nocont_cases.npz         for the spectra of cases.npz: the reference's
                           preprocess_data and fitter_ccf.fit outputs
"""
import os
import sys
import types

os.environ['OMP_NUM_THREADS'] = '1'
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(HERE + '/../..')
sys.path.insert(0, '/tmp/oracle')
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
sys.modules['numba'] = None
sys.modules['numdifftools'] = types.ModuleType('numdifftools')

import numpy as np  # noqa: E402
import make_golden as MG  # noqa: E402  (imports the reference)
from rvspecfit import fitter_ccf, make_ccf, serializer, spec_fit, utils  # noqa


def main():
    if not os.path.exists(MG.TEMPL + 'interp_gold_b.h5'):
        MG.build_reference_artefacts()
    for name, S in MG.SETUPS.items():
        c0, c1, cs = S['ccf']
        make_ccf.main([
            '--setup', name, '--prefix', MG.TEMPL, '--lambda0', str(c0),
            '--lambda1', str(c1), '--step', str(cs), '--every', '20',
            '--vsinis', '0,100', '--oprefix', MG.TEMPL, '--nthreads', '1',
            '--nocontinuum'])
        ci = serializer.load_dict_from_hdf5(
            MG.TEMPL + make_ccf.get_ccf_info_name(name, False))
        cd = np.load(MG.TEMPL + make_ccf.get_ccf_dat_name(name, False))
        cm = np.load(MG.TEMPL + make_ccf.get_ccf_mod_name(name, False))
        cc = ci['ccfconf']
        assert not cc['continuum'] and 'splinestep' not in cc
        vs = np.array([np.nan if _ is None else _ for _ in ci['vsinis']],
                      dtype=float)
        np.savez_compressed(
            HERE + '/lib_nocont_%s.npz' % name,
            ccfnc_fft=cd['fft'], ccfnc_fft2=cd['fft2'], ccfnc_mod=cm,
            ccfnc_params=np.asarray(ci['params'], dtype=float),
            ccfnc_vsinis=vs,
            ccfnc_parnames=np.array(list(ci['parnames'])),
            ccfnc_logl0=np.array(cc['logl0']), ccfnc_logl1=np.array(cc['logl1']),
            ccfnc_npoints=np.array(cc['npoints']),
            ccfnc_continuum=np.array(bool(cc['continuum'])),
            ccfnc_maxcontpts=np.array(cc['maxcontpts']))

    config = dict(utils.read_config(MG.WORK + '/config.yaml'))
    config['ccf_continuum_normalize'] = False
    cases = dict(np.load(HERE + '/cases.npz'))
    R = MG.Rec()
    for t in ('c0', 'c1', 'c2', 'c3'):
        names = [str(_) for _ in cases[t + '/names']]
        sds = [spec_fit.SpecData(n, cases['%s/%s/lam' % (t, n)],
                                 cases['%s/%s/spec' % (t, n)],
                                 cases['%s/%s/espec' % (t, n)],
                                 badmask=cases['%s/%s/badmask' % (t, n)])
               for n in names]
        fitter_ccf.CCFCache.ccfs.clear()
        seen = []
        orig = np.argmin

        def spy(a, *args, **kw):
            seen.append(np.array(a))
            return orig(a, *args, **kw)
        np.argmin = spy
        try:
            res = fitter_ccf.fit(sds, config)
        finally:
            np.argmin = orig
        # fitter_ccf.py:219-221: argmin(all_chisqs.min(axis=1)), argmin(best_ccf)
        tmin = [a for a in seen if a.ndim == 1][0]
        R.put(t + '/template_min', tmin)
        R.put(t + '/best_id', int(orig(tmin)))
        R.put(t + '/best_vel', res['best_vel'])
        R.put(t + '/best_ccf', res['best_ccf'])
        R.put(t + '/best_vsini',
              np.nan if res['best_vsini'] is None else res['best_vsini'])
        R.put(t + '/best_par', np.array([res['best_par'][k] for k in
                                         ('teff', 'logg', 'feh', 'alpha')]))
        for n, sd in zip(names, sds):
            ccfconf = fitter_ccf.get_ccf_info(n, config)[3]['ccfconf']
            ps, pi = make_ccf.preprocess_data(sd.lam, sd.spec, sd.espec,
                                              badmask=sd.badmask,
                                              ccfconf=ccfconf)
            R.put('%s/%s/proc_spec' % (t, n), ps)
            R.put('%s/%s/proc_ivar' % (t, n), pi)
            R.put('%s/%s/best_model' % (t, n), res['best_model'][n])
    np.savez_compressed(HERE + '/nocont_cases.npz', **R.d)
    print('wrote', len(R.d), 'arrays')


if __name__ == '__main__':
    main()
