#!/usr/bin/env python3
"""The reference's accuracy harness (tests/accuracy.py + tests/runall_accuracy.py:
1000 seeded synthetic spectra -- random velocity ~ N(0, 300), random stellar
parameters over the whole grid, a random continuum slope lam**U(-2, 2), a random flux
scale 10**U(-3, 3), Gaussian noise at a given S/N -- each through vel_fit.process from
the same starting point, then the median and scatter of v - v0 and the width of the
pull (v - v0) / vel_err) -- as ONE batch on the GPU instead of a 24-process pool.

    python tools/accuracy_suite.py [S/N = 300] [n = 1000] [library npz = golden 7^4 grid | -] [nobfgs]

prints the reference's two summary lines:
    median(dx) median(err) std(dx) std(dx / err)
    ... the same for the half with the smaller errors
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def make_spectra(lam, n, sn, seed=1, resol=2000.0, lamcen=None):
    """the draws of tests/accuracy.py:doone for n seeds (one RandomState per
    spectrum, seeded like runall_accuracy.py: randint(0, 1e9) of RandomState(seed))"""
    from rvspecfit_amd import synth
    lamcen = 0.5 * (lam[0] + lam[-1]) if lamcen is None else lamcen
    wresol = lamcen / resol / 2.35
    seeds = np.random.RandomState(seed).randint(0, int(1e9), size=n)
    v0 = np.zeros(n)
    truth = np.zeros((n, 4))
    spec = np.zeros((n, len(lam)))
    espec = np.zeros_like(spec)
    c = 299792.458
    for i, s in enumerate(seeds):
        rng = np.random.RandomState(s)
        v0[i] = rng.normal(0, 300)
        slope = rng.uniform(-2, 2)
        teff, feh = rng.uniform(3000, 12000), rng.uniform(-2, 0)
        alpha, logg = rng.uniform(0, 1), rng.uniform(0, 5)
        lam1 = lam / np.sqrt((1 + v0[i] / c) / (1 - v0[i] / c))
        sp0 = synth.spectrum(lam1, teff, logg, feh, alpha, wresol=wresol) * lam**slope
        sp0 = sp0 / np.median(sp0) * 10**rng.uniform(-3, 3)
        espec[i] = sp0 / sn
        spec[i] = rng.normal(sp0, espec[i])
        truth[i] = teff, logg, feh, alpha
    return v0, truth, spec, espec


def run(sn=300.0, n=1000, lib_npz=None, setup='gold_b', lam=None, seed=1, npoly=10,
        config=None):
    """returns dict(v0, vel, vel_err, truth, summary)"""
    import torch
    from rvspecfit_amd import spec_inter, vel_fit
    from rvspecfit_amd.engine import ArmData, SpecBatch
    from rvspecfit_amd.library import TemplateLibrary
    gold = os.path.join(REPO, 'tests', 'golden')
    if lib_npz is None:
        lib_npz = os.path.join(gold, 'lib_%s.npz' % setup)
    root = 'accuracy://' + os.path.basename(lib_npz)
    spec_inter.register_library(TemplateLibrary(setup, dict(np.load(lib_npz))), root)
    if lam is None:
        lam = np.load(os.path.join(gold, 'cases.npz'))['c0/%s/lam' % setup]
    cfg = dict(min_vel=-1500, max_vel=1500, min_vel_step=0.2, vel_step0=5,
               min_vsini=0.1, max_vsini=500, second_minimizer=True)
    # (second_minimizer = True: utils.read_config's default, utils.py:26 -- what the
    # reference's tests/accuracy.py runs; config=dict(second_minimizer=False) for the
    # Nelder-Mead-only run)
    cfg.update(config or {})
    cfg['template_lib'] = root
    v0, truth, spec, espec = make_spectra(lam, n, sn, seed)
    batch = SpecBatch([ArmData(setup, lam, spec, espec)])
    pd0 = dict(logg=np.full(n, 2.5), teff=np.full(n, 5000.), feh=np.full(n, -1.),
               alpha=np.full(n, 0.5))
    res = vel_fit.process(batch, pd0, config=cfg, options=dict(npoly=npoly))
    vel = res['vel'].cpu().numpy()
    err = np.asarray(res['vel_err'].cpu().numpy() if torch.is_tensor(res['vel_err'])
                     else res['vel_err'], dtype=float)
    dx = vel - v0
    half = err < np.median(err)
    summary = dict(median_dx=float(np.median(dx)), median_err=float(np.median(err)),
                   std_dx=float(np.std(dx)), std_pull=float(np.std(dx / err)),
                   median_dx_half=float(np.median(dx[half])),
                   median_err_half=float(np.median(err[half])),
                   std_dx_half=float(np.std(dx[half])))
    return dict(v0=v0, vel=vel, vel_err=err, truth=truth, summary=summary, res=res)


if __name__ == '__main__':
    sn = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    import time
    t0 = time.time()
    out = run(sn, n, sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != '-' else None,
              config=dict(second_minimizer=False) if 'nobfgs' in sys.argv[3:] else None)
    s = out['summary']
    # (the two lines runall_accuracy.py prints)
    print(s['median_dx'], s['median_err'], s['std_dx'], s['std_pull'])
    print(s['median_dx_half'], s['median_err_half'], s['std_dx_half'])
    dx, err = out['vel'] - out['v0'], out['vel_err']
    pull = dx / err
    q = np.percentile(pull, [16, 50, 84])
    print('n %d S/N %g: pull percentiles 16/50/84 %.3f %.3f %.3f, |pull| > 5: %d, '
          '|dx| > 50 km/s: %d, %.1f s' % (n, sn, q[0], q[1], q[2],
                                         int((np.abs(pull) > 5).sum()),
                                         int((np.abs(dx) > 50).sum()), time.time() - t0))
