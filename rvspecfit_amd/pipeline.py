"""The contract unit of work for a batch of spectra:

    fitter_ccf.fit  ->  template build at the CCF best_par / best_vsini
    ->  find_best over arange(min_vel, max_vel, vel_step0) x 1 template
    ->  get_chisq_continuum            [-> optional _minimum_sampler refinement]

which is what desi_fit.proc_onespec (desi/desi_fit.py:288-309) runs per fibre
around the optimiser.  Everything stays in HBM; the result is one fixed-size
float64 record per spectrum (RECORD_FIELDS) so that multi-GPU runs gather a
[S, NREC] tensor and nothing else.
"""
import numpy as np
import torch

from . import engine
from . import spec_inter
from . import spec_fit
from . import vel_fit

RECORD_FIELDS = ('best_id', 'vrad_ccf', 'p0', 'p1', 'p2', 'p3', 'vsini',
                 'best_vel', 'vel_err', 'skewness', 'kurtosis', 'best_chi',
                 'chisq_c0', 'chisq_c1', 'chisq_c2', 'status')
NREC = len(RECORD_FIELDS)


def fit_batch(batch, config, options=None, refine=False, timers=None):
    """Returns rec float64 [S, NREC] (device).  `timers`, if a dict, receives
    per-stage (start, end) torch.cuda events."""
    options = options or {}
    npoly = options.get('npoly') or 5
    rbf = options.get('rbf_continuum', True)
    S, dev = batch.S, batch.device
    libs = spec_inter.get_libs(batch.names, config)
    ev = _Ev(timers)

    ev.start('ccf')
    ccf = engine.ccf_fit(batch, libs, config)
    ev.stop('ccf')
    ref = libs[batch.names[0]].ccf
    params = ref['params_dev'][ccf['best_id']].contiguous()
    vsini = ref['vsinis_dev'][ccf['best_id']].contiguous()

    ev.start('template')
    coefs, outs = [], []
    for arm in batch.arms:
        c, o = engine.build_templates(libs[arm.name], params, vsini)
        coefs.append(c)
        outs.append(o)
    ev.stop('template')

    ev.start('chisq_grid')
    vg = torch.as_tensor(
        np.arange(config['min_vel'], config['max_vel'],
                  config['vel_step0']).astype(np.float64)).to(dev)
    chisq, status = engine.chisq_grid(
        batch, libs, coefs, outs, vg, npoly=npoly, rbf=rbf,
        vel_bounds=(float(config['min_vel']), float(config['max_vel'])))
    res, _, mst = engine.grid_moments(chisq, vg, Np=1)
    ev.stop('chisq_grid')

    ev.start('continuum')
    cont = engine.chisq_continuum(batch, npoly=npoly, rbf=rbf)
    ev.stop('continuum')

    rec = torch.zeros((S, NREC), dtype=torch.float64, device=dev)
    rec[:, 0] = ccf['best_id'].double()
    rec[:, 1] = ccf['best_vel']
    nd = params.shape[1]
    rec[:, 2:2 + min(nd, 4)] = params[:, :4]
    rec[:, 6] = vsini
    rec[:, 7] = res[:, 1]
    rec[:, 8] = res[:, 2]
    rec[:, 9] = res[:, 4]
    rec[:, 10] = res[:, 3]
    rec[:, 11] = res[:, 0]
    for ia in range(min(3, len(batch.arms))):
        rec[:, 12 + ia] = cont[ia]['true_chisq']
    cst = cont[0]['status']
    for c in cont[1:]:
        cst = cst | c['status']
    rec[:, 15] = (status | mst | ccf['status'] | cst).double()

    if refine:
        ev.start('refine')
        r = vel_fit._minimum_sampler_batch(batch, res[:, 1].cpu().numpy(), params,
                                           vsini, config, options)
        rec[:, 7] = torch.as_tensor(r['best_vel']).to(dev)
        rec[:, 8] = torch.as_tensor(r['vel_err']).to(dev)
        rec[:, 9] = torch.as_tensor(r['skewness']).to(dev)
        rec[:, 10] = torch.as_tensor(r['kurtosis']).to(dev)
        ev.stop('refine')
    return rec


class _Ev:

    def __init__(self, timers):
        self.t = timers

    def start(self, k):
        if self.t is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.t[k] = [e, None]

    def stop(self, k):
        if self.t is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.t[k][1] = e
