"""chi^2-grid callers -- API mirror of the grid-driven parts of
py/rvspecfit/vel_fit.py: firstguess (:13-94), _minimum_sampler (:358-439),
_find_best_vel_iterate (:315-355) and the first step of process (:571-602).

The optimiser / Hessian stage of `process` (Nelder-Mead -> BFGS ->
numdifftools) is outside the accelerated hot path.
"""
import itertools
import logging
import math

import numpy as np
import torch

from . import engine
from . import spec_fit
from . import spec_inter
from .spec_fit import as_batch


def firstguess(specdata, options=None, config=None, resolParams=None,
               vsinigrid=(None, 10, 100), paramsgrid=None):
    """vel_fit.firstguess (vel_fit.py:13-94).  One spectrum -> dict of best
    parameters; SpecBatch -> dict of [S] tensors."""
    if resolParams is not None:
        raise NotImplementedError('resolution matrices: SURVEY 8(f) rank 4')
    options = options or {}
    if paramsgrid is None:
        paramsgrid = {'logg': [1, 2, 3, 4, 5], 'teff': [3000, 5000, 8000, 10000],
                      'feh': [-2, -1, 0], 'alpha': [0]}
    batch, is_batch = as_batch(specdata)
    names = spec_inter.getSpecParams(batch.names[0], config)
    params = []
    for x in itertools.product(*paramsgrid.values()):
        d = dict(zip(paramsgrid.keys(), x))
        params.append([d[_] for _ in names])
    vg = np.arange(config['min_vel'], config['max_vel'], config['vel_step0'])
    S, dev = batch.S, batch.device
    best_chi = torch.full((S, ), float('inf'), dtype=torch.float64, device=dev)
    best_par = torch.zeros((S, len(names)), dtype=torch.float64, device=dev)
    best_vs = torch.full((S, ), float('nan'), dtype=torch.float64, device=dev)
    for vs in vsinigrid:
        rot = None if vs is None else (vs, )
        r = spec_fit.find_best(batch, vg, params, rot_params=rot,
                               config=config, options=options)
        better = r['best_chi'] < best_chi
        best_chi = torch.where(better, r['best_chi'], best_chi)
        best_par = torch.where(better[:, None], r['best_param'], best_par)
        if vs is not None:
            best_vs = torch.where(better, torch.full_like(best_vs, float(vs)),
                                  best_vs)
        else:
            best_vs = torch.where(better, torch.full_like(best_vs, float('nan')),
                                  best_vs)
    if is_batch:
        out = {k: best_par[:, i] for i, k in enumerate(names)}
        out['vsini'] = best_vs
        return out
    bp = best_par[0].cpu().numpy()
    out = {}
    for i, k in enumerate(names):
        v = float(bp[i])
        out[k] = int(v) if v == int(v) else v
    v = float(best_vs[0].item())
    if not np.isnan(v):
        out['vsini'] = int(v) if v == int(v) else v
    return out


def _minimum_sampler_batch(batch, best_vel, best_param, vsini, config, options,
                           crit_ratio=5, goal_width=10, max_points=2048):
    """Batched _minimum_sampler (vel_fit.py:358-439): every spectrum carries
    its own (min_vel, max_vel, step) state; per round all spectra that are not
    converged are evaluated on their own velocity grids in one launch set.
    Grid construction follows the reference formula exactly:
        arange(ceil((min_vel-best_vel)/step)*step, max_vel-best_vel, step)+best_vel
    (host float64, per spectrum).  Returns per-spectrum numpy arrays."""
    S, dev = batch.S, batch.device
    min_vel = np.full(S, float(config['min_vel']))
    max_vel = np.full(S, float(config['max_vel']))
    step = np.full(S, float(config['vel_step0']))
    min_vel_step = config['min_vel_step']
    bv = np.clip(np.asarray(best_vel, dtype=np.float64), min_vel, max_vel)
    err = np.zeros(S)
    skw = np.zeros(S)
    kur = np.zeros(S)
    active = np.ones(S, dtype=bool)
    ngrids = np.zeros(S, dtype=int)
    ngrid_pts = np.zeros(S, dtype=int)
    params = best_param if isinstance(best_param, torch.Tensor) else \
        torch.as_tensor(np.asarray(best_param, dtype=np.float64)).to(dev)
    if params.dim() == 1:
        params = params[None].expand(S, -1)
    all_grids = [[] for _ in range(S)]
    for it in range(10):
        idx = np.nonzero(active)[0]
        if len(idx) == 0:
            break
        grids = []
        for i in idx:
            g = np.arange(math.ceil((min_vel[i] - bv[i]) / step[i]) * step[i],
                          max_vel[i] - bv[i], step[i]) + bv[i]
            grids.append(g)
            all_grids[i].append(g)
        nmax = max(len(g) for g in grids)
        if nmax > max_points:
            raise RuntimeError('velocity grid too long')
        vg = np.zeros((len(idx), nmax))
        nv = np.zeros(len(idx), dtype=np.int32)
        for k, g in enumerate(grids):
            vg[k, :len(g)] = g
            vg[k, len(g):] = g[-1]  # padding, ignored through nvel
            nv[k] = len(g)
        sub = _sub_batch(batch, idx)
        vgt = torch.as_tensor(vg).to(dev)
        idt = torch.as_tensor(idx).to(dev)
        p = params[idt][:, None, :].contiguous()
        vs = None if vsini is None else vsini[idt]
        chisq, status, _ = spec_fit.chisq_grid_jobs(sub, vgt, p, vs, options,
                                                    config)
        res, _, _ = engine.grid_moments(chisq.reshape(len(idx), -1), vgt, Np=1,
                                        nvel=torch.as_tensor(nv).to(dev))
        r = res.cpu().numpy()
        for k, i in enumerate(idx):
            bv[i], err[i], kur[i], skw[i] = r[k, 1], r[k, 2], r[k, 3], r[k, 4]
            ngrids[i] += 1
            ngrid_pts[i] += nv[k]
            if step[i] < err[i] / crit_ratio or step[i] < min_vel_step:
                active[i] = False
                continue
            if step[i] > err[i]:
                new_step, width = step[i] / crit_ratio, step[i] * goal_width
            else:
                new_step, width = err[i] / crit_ratio * 0.8, err[i] * goal_width
            min_vel[i] = max(bv[i] - width, min_vel[i])
            max_vel[i] = min(bv[i] + width, max_vel[i])
            step[i] = new_step
    return dict(best_vel=bv, vel_err=err, skewness=skw, kurtosis=kur,
                ngrids=ngrids, npoints=ngrid_pts, grids=all_grids)


_sub_cache = {}


def _sub_batch(batch, idx):
    """view of a subset of spectra as a SpecBatch (full batch -> itself)"""
    if len(idx) == batch.S:
        return batch
    idt = torch.as_tensor(idx).to(batch.device)
    arms = [engine.ArmData(a.name, a.lam_host, a.spec[idt], a.espec[idt],
                           a.badmask[idt], device=batch.device)
            for a in batch.arms]
    return engine.SpecBatch(arms)


def _find_best_vel_iterate(best_vel, min_vel, max_vel, vel_step0, specdata=None,
                           best_param=None, resolParams=None, config=None,
                           options=None, min_vel_step=None):
    """vel_fit._find_best_vel_iterate (vel_fit.py:315-355) for one spectrum."""
    if best_vel > max_vel or best_vel < min_vel:
        logging.warning('Velocity too large...')
    batch, _ = as_batch(specdata)
    rot = best_param['rot_params']
    vs = None
    if rot is not None:
        vs = torch.as_tensor(np.asarray(rot, dtype=np.float64)).to(batch.device)
    cfg = dict(config)
    cfg.update(min_vel=min_vel, max_vel=max_vel, vel_step0=vel_step0,
               min_vel_step=min_vel_step)
    r = _minimum_sampler_batch(batch, [best_vel], best_param['params'], vs, cfg,
                               options)
    return (float(r['best_vel'][0]), float(r['vel_err'][0]),
            float(r['skewness'][0]), float(r['kurtosis'][0]))


def process(specdata, paramDict0, fixParam=None, options=None, config=None,
            resolParams=None, priors=None):
    """Only the grid-driven first stage of vel_fit.process is accelerated
    (vel_fit.py:571-602: find_best over arange(min_vel, max_vel, vel_step0) at
    the starting parameters) followed by the velocity refinement
    (vel_fit.py:672) and the full-output evaluation (:689).  The Nelder-Mead /
    BFGS / Hessian stage is SURVEY 8(f) rank 1 and is NOT run: parameters are
    returned as given."""
    if config is None:
        raise RuntimeError('Config must be provided')
    options = options or {}
    batch, is_batch = as_batch(specdata)
    if is_batch:
        raise NotImplementedError('use pipeline.fit_batch for batches')
    names = spec_inter.getSpecParams(batch.names[0], config)
    curparam = tuple(paramDict0[_] for _ in names)
    rot = (paramDict0['vsini'], ) if 'vsini' in paramDict0 else None
    vg = np.arange(config['min_vel'], config['max_vel'], config['vel_step0'])
    res = spec_fit.find_best(specdata, vg, [curparam], rot_params=rot,
                             config=config, options=options)
    bv, be, sk, ku = _find_best_vel_iterate(
        res['best_vel'], config['min_vel'], config['max_vel'],
        config['vel_step0'], specdata=specdata,
        best_param=dict(params=curparam, rot_params=rot), config=config,
        options=options, min_vel_step=config['min_vel_step'])
    outp = spec_fit.get_chisq(specdata, bv, curparam, rot, options=options,
                              config=config, full_output=True)
    ret = dict(param=dict(zip(names, curparam)), vel=bv, vel_err=be,
               vel_skewness=sk, vel_kurtosis=ku, yfit=outp['models'],
               raw_models=outp['raw_models'], chisq=outp['chisq'],
               logl=outp['logl'], chisq_array=outp['chisq_array'],
               npix_array=outp['npix_array'], minimize_success=False,
               optimizer_run=False)
    if rot is not None:
        ret['vsini'] = rot[0]
    return ret
