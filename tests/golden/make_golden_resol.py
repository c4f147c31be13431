#!/opt/conda/bin/python3.9
"""Golden vectors for the resolution-matrix path, generated
by IMPORTING the reference (build container only; run make_golden.py first):

    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_resol.py

Writes tests/golden/resol_cases.npz: the reference's construct_resol_mat
matrices (dia storage) and its get_chisq / find_best / get_chisq_continuum
outputs with `resol_params` (one matrix per setup) and with per-SpecData
`resolution` matrices, on the seeded spectra of cases.npz.
"""
import os
import sys

os.environ['OMP_NUM_THREADS'] = '1'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402
import scipy.sparse  # noqa: E402
import make_golden as mg  # noqa: E402  (sets up the reference import)
from rvspecfit import spec_fit, utils  # noqa: E402

SPEC = {
    'c1': dict(names=['gold_b', 'gold_r'], truth=(6123., 2.5, -0.4, 0.1),
               vel=-212.7, snr=30., seed=102, mask=0.05, slope=0.3),
    'c2': dict(names=['gold_r'], truth=(4200., 3.5, -1.7, 0.3), vel=402.1,
               snr=1000., seed=103, mask=0.0, slope=-0.2),
}


def dia_arrays(M):
    D = scipy.sparse.dia_matrix(M)
    return np.asarray(D.data), np.asarray(D.offsets)


def main():
    config = utils.read_config(mg.WORK + '/config.yaml')
    out = {}
    vel_grid = np.arange(-300., 300., 5.)
    out['vel_grid'] = vel_grid
    for tag, s in SPEC.items():
        sds, raw = mg.make_specdata(s['names'], s['truth'], s['vel'], s['snr'],
                                    s['seed'], s['mask'], s['slope'])
        # (a) resol_params: one matrix per setup from a resolving power
        rp = {}
        for n, (lam, spec, espec, bm) in zip(s['names'], raw):
            R = spec_fit.construct_resol_mat(lam, resol=2500.)
            rp[n] = R
            d, o = dia_arrays(R.mat)
            out['%s/rp/%s/data' % (tag, n)] = d
            out['%s/rp/%s/offsets' % (tag, n)] = o
        opt = dict(npoly=10)
        trials = [(s['vel'], s['truth'], None), (s['vel'] + 7.5, s['truth'],
                                                 (30., )),
                  (-120.3, (5000., 2., -1., 0.2), None)]
        for i, (v, p, rot) in enumerate(trials):
            val = spec_fit.get_chisq(sds, v, p, rot, rp, options=opt,
                                     config=config)
            out['%s/rp/t%d/vel' % (tag, i)] = np.array(v)
            out['%s/rp/t%d/param' % (tag, i)] = np.array(p)
            out['%s/rp/t%d/vsini' % (tag, i)] = np.array(
                np.nan if rot is None else rot[0])
            out['%s/rp/t%d/value' % (tag, i)] = np.array(val)
        full = spec_fit.get_chisq(sds, s['vel'], s['truth'], None, rp,
                                  options=opt, config=config, full_output=True)
        out[tag + '/rp/full/chisq'] = np.array(full['chisq'])
        out[tag + '/rp/full/chisq_array'] = np.array(full['chisq_array'])
        for n, m, rm in zip(s['names'], full['models'], full['raw_models']):
            out['%s/rp/full/model_%s' % (tag, n)] = m
            out['%s/rp/full/raw_model_%s' % (tag, n)] = rm
        fb = spec_fit.find_best(sds, vel_grid, [s['truth']], None, rp,
                                options=opt, config=config)
        out[tag + '/rp/find_best'] = np.array([fb['best_vel'], fb['vel_err'],
                                               fb['best_chi']])
        grid = np.array([spec_fit.get_chisq(sds, v, s['truth'], None, rp,
                                            options=opt, config=config)
                         for v in vel_grid])
        out[tag + '/rp/grid'] = grid
        # (b) per-spectrum resolution matrices (DESI style), narrower kernel
        sds2 = []
        for n, (lam, spec, espec, bm) in zip(s['names'], raw):
            R = spec_fit.construct_resol_mat(lam, width=0.55)
            d, o = dia_arrays(R.mat)
            out['%s/own/%s/data' % (tag, n)] = d
            out['%s/own/%s/offsets' % (tag, n)] = o
            sds2.append(spec_fit.SpecData(n, lam, spec, espec, badmask=bm,
                                          resolution=R))
        for i, (v, p, rot) in enumerate(trials):
            val = spec_fit.get_chisq(sds2, v, p, rot, options=opt, config=config)
            out['%s/own/t%d/value' % (tag, i)] = np.array(val)
        c = spec_fit.get_chisq_continuum(sds2, options=opt)
        out[tag + '/own/cont/chisq_array'] = c['chisq_array']
        out[tag + '/own/cont/redchisq_array'] = c['redchisq_array']
        fb = spec_fit.find_best(sds2, vel_grid, [s['truth']], (30., ),
                                options=opt, config=config)
        out[tag + '/own/find_best'] = np.array([fb['best_vel'], fb['vel_err'],
                                                fb['best_chi']])
        # (c) A7 fast_interp (nearest knot) and a per-setup espec_systematic dict
        for i, (v, p, rot) in enumerate(trials):
            out['%s/fast/t%d/value' % (tag, i)] = np.array(spec_fit.get_chisq(
                sds, v, p, rot, options=opt, config=config, fast_interp=True))
        esd = {n: 0.02 * (1 + k) for k, n in enumerate(s['names'])}
        out[tag + '/esys_dict/vals'] = np.array([esd[n] for n in s['names']])
        out[tag + '/esys_dict/value'] = np.array(spec_fit.get_chisq(
            sds, s['vel'], s['truth'], None, options=opt, config=config,
            espec_systematic=esd))
        full = spec_fit.get_chisq(sds, s['vel'], s['truth'], None, options=opt,
                                  config=config, fast_interp=True,
                                  full_output=True)
        out[tag + '/fast/full/chisq_array'] = np.array(full['chisq_array'])
        print(tag, 'rp', [float(out['%s/rp/t%d/value' % (tag, i)])
                          for i in range(3)], 'own',
              [float(out['%s/own/t%d/value' % (tag, i)]) for i in range(3)],
              out[tag + '/rp/find_best'], out[tag + '/own/find_best'],
              {n: out['%s/rp/%s/offsets' % (tag, n)].shape for n in s['names']})
    np.savez_compressed(HERE + '/resol_cases.npz', **out)
    print('wrote', os.path.getsize(HERE + '/resol_cases.npz'))


if __name__ == '__main__':
    main()
