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


def fit_batch(batch, config, options=None, refine=False, timers=None,
              share_templates=True):
    """Returns rec float64 [S, NREC] (device).  `timers`, if a dict, receives
    per-stage (start, end) torch.cuda events.  share_templates=False builds one
    template per spectrum even when spectra share a CCF node (same records, bit
    for bit; tests and the bench's --per-spectrum-templates line)."""
    options = options or {}
    npoly = options.get('npoly') or 5
    rbf = options.get('rbf_continuum', True)
    S, dev = batch.S, batch.device
    libs = spec_inter.get_libs(batch.names, config)
    ev = _Ev(timers)

    ev.start('ccf')
    ccf = engine.ccf_fit(batch, libs, config)
    ev.stop('ccf')
    ref = libs[batch.names[0]].ccf_set(config)
    params = ref['params_dev'][ccf['best_id']].contiguous()
    vsini = ref['vsinis_dev'][ccf['best_id']].contiguous()

    ev.start('template')
    # The template of a spectrum is the one of its CCF node (best_par, best_vsini):
    # a batch larger than the CCF set meets every node many times, so each node's
    # template (interpolation, broadening, spline records) is built ONCE and the
    # chi^2 kernels are pointed at it per job -- what the reference's caches do
    # for successive spectra (getCurTempl's lru_cache(100) on the parameter
    # tuple, the spline cache on templ_tag: spec_fit.py:357-407, 902-910).  Same
    # inputs to the same kernels: the records a spectrum sees are bit for bit
    # those of a template built for it alone.
    Tn = ref['T']
    # (while a node set's spline records stay cache resident: 76 nodes x 200 KB =
    # 15 MB per arm; at 534 nodes -- 107 MB per arm -- the chi^2 kernel's gathers
    # miss as often as with per-spectrum templates, whose XCD placement fetches a
    # job's records once: 101.7 against 96.7 ms per step, a wash with the 4 ms the
    # template stage saves.  Jobs sorted by node so that an XCD holds one node at
    # a time were measured too: every CU then hammers the same lines, 94.5
    # against 89 ms at 76 nodes.)
    max_ntp = max(libs[a.name].ntp for a in batch.arms)
    shared = share_templates and Tn <= S and Tn * max_ntp * 32 <= (64 << 20)
    tparams = ref['params_dev'] if shared else params
    tvsini = ref['vsinis_dev'] if shared else vsini
    templ_rows = ccf['best_id'].to(torch.int32).contiguous() if shared else None
    coefs, outs = [], []
    for arm in batch.arms:
        c, o = engine.build_templates(libs[arm.name], tparams, tvsini)
        coefs.append(c)
        outs.append(o)
    ev.stop('template')

    ev.start('chisq_grid')
    vg = torch.as_tensor(
        np.arange(config['min_vel'], config['max_vel'],
                  config['vel_step0']).astype(np.float64)).to(dev)
    # (the redo of ill-conditioned jobs -- a host look at the status vector -- is
    # deferred behind everything that can be queued without it: at 5 ms per step
    # (BASELINE configs[1]) a synchronisation in the middle of the step showed)
    chisq, status, grid_finish = engine.chisq_grid(
        batch, libs, coefs, outs, vg, npoly=npoly, rbf=rbf, job_templ=templ_rows,
        vel_bounds=(float(config['min_vel']), float(config['max_vel'])),
        defer_redo=True)
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
    for ia in range(min(3, len(batch.arms))):
        rec[:, 12 + ia] = cont[ia]['true_chisq']
    cst = cont[0]['status']
    for c in cont[1:]:
        cst = cst | c['status']

    def grid_fields(res, mst):
        rec[:, 7] = res[:, 1]
        rec[:, 8] = res[:, 2]
        rec[:, 9] = res[:, 4]
        rec[:, 10] = res[:, 3]
        rec[:, 11] = res[:, 0]
        rec[:, 15] = (status | mst | ccf['status'] | cst).double()
    grid_fields(res, mst)
    if grid_finish():   # (rare) some jobs went through the point kernel
        res, _, mst = engine.grid_moments(chisq, vg, Np=1)
        grid_fields(res, mst)

    if refine:
        ev.start('refine')
        r = vel_fit._minimum_sampler_batch(batch, res[:, 1].contiguous(), params,
                                           vsini, config, options,
                                           templates=(coefs, outs),
                                           templ_rows=templ_rows)
        rec[:, 7] = torch.as_tensor(r['best_vel']).to(dev)
        rec[:, 8] = torch.as_tensor(r['vel_err']).to(dev)
        rec[:, 9] = torch.as_tensor(r['skewness']).to(dev)
        rec[:, 10] = torch.as_tensor(r['kurtosis']).to(dev)
        ev.stop('refine')
    return rec


PROCESS_FIELDS = ('vel', 'vel_err', 'vel_skewness', 'vel_kurtosis', 'vsini',
                  'teff', 'logg', 'feh', 'alpha', 'teff_err', 'logg_err',
                  'feh_err', 'alpha_err', 'chisq', 'minimize_success',
                  'bad_hessian', 'status')
NPROC = len(PROCESS_FIELDS)


def process_batch(batch, rec, config, options=None, fixParam=None, priors=None):
    """The optimiser stage (vel_fit.process, as desi_fit.py:289-309 calls it)
    for a batch, started from the CCF parameters of `rec` (fit_batch).  Returns
    a fixed-size float64 record [S, NPROC] (PROCESS_FIELDS) so that multi-GPU
    runs gather one tensor, like fit_batch."""
    options = options or {}
    S, dev = batch.S, batch.device
    names = spec_inter.getSpecParams(batch.names[0], config)
    pd0 = {k: rec[:, 2 + i].contiguous() for i, k in enumerate(names)}
    vs = rec[:, 6]
    pd0['vsini'] = torch.where(torch.isfinite(vs), vs,
                               torch.zeros_like(vs)).contiguous()
    r = vel_fit.process(batch, pd0, fixParam=fixParam, options=options,
                        config=config, priors=priors)
    out = torch.full((S, NPROC), float('nan'), dtype=torch.float64, device=dev)
    out[:, 0] = r['vel']
    out[:, 1] = r['vel_err']
    out[:, 2] = r['vel_skewness']
    out[:, 3] = r['vel_kurtosis']
    if 'vsini' in r:
        out[:, 4] = r['vsini']
    for i, k in enumerate(('teff', 'logg', 'feh', 'alpha')):
        if k in r['param']:
            out[:, 5 + i] = r['param'][k]
            out[:, 9 + i] = torch.as_tensor(r['param_err'][k]).to(dev)
    out[:, 13] = r['chisq']
    out[:, 14] = r['minimize_success'].double()
    out[:, 15] = torch.as_tensor(r['bad_hessian'].astype(np.float64)).to(dev)
    out[:, 16] = r['status'].double()
    return out


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
