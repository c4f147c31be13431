"""chi^2-grid callers -- API mirror of the grid-driven parts of
py/rvspecfit/vel_fit.py: firstguess (:13-94), _minimum_sampler (:358-439),
_find_best_vel_iterate (:315-355) and the first step of process (:571-602).

`process` (vel_fit.py:505-737) runs its Nelder-Mead stage as S lock-step
simplices (optimizer.py over csrc/nm.hip) over batched objective evaluations (SURVEY 8(f)
rank 1); see its docstring for what differs from the reference.
"""
import contextlib
import itertools
import logging
import math
import os
import threading
import time
import types

import numpy as np
import torch

from . import engine
from . import numdiff
from . import spec_fit
from . import spec_inter
from .engine import SpecBatch
from .spec_fit import as_batch


def firstguess(specdata, options=None, config=None, resolParams=None,
               vsinigrid=(None, 10, 100), paramsgrid=None):
    """vel_fit.firstguess (vel_fit.py:13-94).  One spectrum -> dict of best
    parameters; SpecBatch -> dict of [S] tensors."""
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
                               resol_params=resolParams, config=config,
                               options=options)
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
                           crit_ratio=5, goal_width=10, grid_budget=1 << 24,
                           resolParams=None, keep_grids=False, templates=None,
                           templ_rows=None):
    """Batched _minimum_sampler (vel_fit.py:358-439): every spectrum carries
    its own (min_vel, max_vel, step) state; per round all spectra that are not
    converged are evaluated on their own velocity grids in one launch set.
    Grid construction follows the reference formula exactly:
        arange(ceil((min_vel-best_vel)/step)*step, max_vel-best_vel, step)+best_vel
    (host float64, per spectrum).  Returns per-spectrum numpy arrays.
    The templates (spline records of every spectrum's parameters, all arms) are
    built once -- or handed over by the caller as `templates` = (coefs, outs) of
    engine.build_templates, with `templ_rows` (int32 [S]) naming the row of each
    spectrum when spectra share templates -- and every round's grids are evaluated against
    them: the reference's get_chisq finds them in its template cache the same way
    (spec_fit.py:902-910)."""
    S, dev = batch.S, batch.device
    opts = options or {}
    npoly = opts.get('npoly') or 5
    rbf = opts.get('rbf_continuum', True)
    libs = spec_inter.get_libs(batch.names, config)
    # The per-spectrum state (window, step, current minimum) lives on the device
    # and every formula below is the reference's, evaluated element-wise in
    # float64 by separate (un-fused) operations -- the same roundings as numpy's:
    #     arange(ceil((min_vel-best_vel)/step)*step, max_vel-best_vel, step)+best_vel
    # The host looks at two numbers per round (how many spectra are still
    # active, the longest grid); no grid ever crosses PCIe.
    f64 = dict(dtype=torch.float64, device=dev)
    min_vel = torch.full((S, ), float(config['min_vel']), **f64)
    max_vel = torch.full((S, ), float(config['max_vel']), **f64)
    step = torch.full((S, ), float(config['vel_step0']), **f64)
    min_vel_step = config['min_vel_step']
    bv0 = best_vel if isinstance(best_vel, torch.Tensor) else \
        torch.as_tensor(np.asarray(best_vel, dtype=np.float64))
    bv = torch.minimum(torch.maximum(bv0.to(**f64).reshape(-1), min_vel),
                       max_vel).clone()
    err = torch.zeros(S, **f64)
    skw = torch.zeros(S, **f64)
    kur = torch.zeros(S, **f64)
    active = torch.ones(S, dtype=torch.bool, device=dev)
    ngrids = torch.zeros(S, dtype=torch.int64, device=dev)
    ngrid_pts = torch.zeros(S, dtype=torch.int64, device=dev)
    params = best_param if isinstance(best_param, torch.Tensor) else \
        torch.as_tensor(np.asarray(best_param, dtype=np.float64)).to(dev)
    if params.dim() == 1:
        params = params[None].expand(S, -1)
    all_grids = [[] for _ in range(S)] if keep_grids else None
    if templates is None:
        coefs, outs = [], []
        for arm in batch.arms:
            c, o = engine.build_templates(libs[arm.name], params.contiguous(),
                                          vsini)
            coefs.append(c)
            outs.append(o)
    else:
        coefs, outs = templates
    resols = spec_fit._resols(batch, resolParams)
    nanv = torch.full((1, ), float('nan'), **f64)
    # (a tensor divisor: torch turns `x / python_scalar` into a multiplication by
    # the reciprocal on the GPU, one ulp away from numpy's division)
    crit = torch.full((1, ), float(crit_ratio), **f64)
    for it in range(10):
        idx = torch.nonzero(active).reshape(-1)
        n = int(idx.numel())
        if n == 0:
            break
        st_, bv_ = step[idx], bv[idx]
        start = torch.ceil((min_vel[idx] - bv_) / st_) * st_
        stop = max_vel[idx] - bv_
        nv = torch.clamp(torch.ceil((stop - start) / st_), min=0).to(torch.int64)
        nvmax = int(nv.max().item())
        # the reference accepts any grid length ((max_vel - min_vel) / vel_step0
        # is the user's choice): long grids are evaluated for fewer spectra at a
        # time, so that a launch set holds at most grid_budget velocities
        r = torch.empty((n, 8), **f64)
        rows = max(1, int(grid_budget // max(nvmax, 1)))
        for a in range(0, n, rows):
            sl = slice(a, a + rows)
            nvc = nv[sl]
            nmax = max(nvmax if rows >= n else int(nvc.max().item()), 1)
            ii = torch.arange(nmax, **f64)
            vg = (start[sl, None] + ii[None, :] * st_[sl, None]) + bv_[sl, None]
            last = vg.gather(1, torch.clamp(nvc - 1, min=0)[:, None])
            vg = torch.where(ii[None, :] < nvc[:, None], vg, last).contiguous()
            if keep_grids:
                vh, nh = vg.cpu().numpy(), nvc.cpu().numpy()
                for k, i in enumerate(idx[sl].cpu().numpy()):
                    all_grids[i].append(vh[k, :nh[k]].copy())
            idt = None if (rows >= n and n == S) else \
                idx[sl].to(torch.int32).contiguous()
            jt = idt
            if templ_rows is not None:
                jt = templ_rows if idt is None else \
                    templ_rows[idx[sl]].contiguous()
            chisq, status = engine.chisq_grid(
                batch, libs, coefs, outs, vg, npoly=npoly, rbf=rbf,
                job_spec=idt, job_templ=jt, resols=resols)
            res, _, _ = engine.grid_moments(
                chisq.reshape(nvc.shape[0], -1), vg, Np=1,
                nvel=nvc.to(torch.int32).contiguous())
            r[sl] = res
        # a spectrum whose grid has no finite minimum (every chi^2 non finite,
        # an empty grid) leaves the loop with NaN results instead of steering
        # the next grid of the whole batch with them
        lost = ~(torch.isfinite(r[:, 1]) & torch.isfinite(r[:, 2])) | (nv < 1)
        keep = ~lost
        bv[idx] = torch.where(lost, nanv, r[:, 1])
        err[idx] = torch.where(lost, nanv, r[:, 2])
        kur[idx] = torch.where(lost, nanv, r[:, 3])
        skw[idx] = torch.where(lost, nanv, r[:, 4])
        ngrids[idx] += keep.to(torch.int64)
        ngrid_pts[idx] += torch.where(keep, nv, torch.zeros_like(nv))
        e_ = err[idx]
        done = (st_ < e_ / crit) | (st_ < min_vel_step)
        coarse = st_ > e_
        new_step = torch.where(coarse, st_ / crit, e_ / crit * 0.8)
        width = torch.where(coarse, st_ * goal_width, e_ * goal_width)
        go = keep & ~done
        active[idx] = go
        bvn = bv[idx]
        min_vel[idx] = torch.where(go, torch.maximum(bvn - width, min_vel[idx]),
                                   min_vel[idx])
        max_vel[idx] = torch.where(go, torch.minimum(bvn + width, max_vel[idx]),
                                   max_vel[idx])
        step[idx] = torch.where(go, new_step, st_)
    bv, err, skw, kur = (_.cpu().numpy() for _ in (bv, err, skw, kur))
    ngrids, ngrid_pts = ngrids.cpu().numpy(), ngrid_pts.cpu().numpy()
    return dict(best_vel=bv, vel_err=err, skewness=skw, kurtosis=kur,
                ngrids=ngrids, npoints=ngrid_pts, grids=all_grids)


_sub_cache = {}


def _sub_batch(batch, idx):
    """view of a subset of spectra as a SpecBatch (full batch -> itself)"""
    if len(idx) == batch.S:
        return batch
    return batch.subset(torch.as_tensor(idx).to(batch.device))


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
                               options, resolParams=resolParams)
    return (float(r['best_vel'][0]), float(r['vel_err'][0]),
            float(r['skewness'][0]), float(r['kurtosis'][0]))


class VSiniMapper:
    """vel_fit.VSiniMapper (vel_fit.py:95-116) on tensors"""

    def __init__(self, max_vsini):
        self.max_vsini = max_vsini

    def to_internal(self, vsini):
        if not isinstance(vsini, torch.Tensor):   # (one spectrum: the reference's form)
            return np.clip(vsini, 0, self.max_vsini)
        return torch.clamp(vsini, 0, self.max_vsini)

    def to_vsini(self, x):
        if not isinstance(x, torch.Tensor):
            vsini = np.clip(x, 0, self.max_vsini)
            penalty = int(x < 0) * (vsini - x)**2 + int(x > self.max_vsini) * \
                (vsini - x)**2
            return vsini, penalty
        vsini = torch.clamp(x, 0, self.max_vsini)
        out = (x < 0) | (x > self.max_vsini)
        penalty = torch.where(out, (vsini - x)**2, torch.zeros_like(x))
        return vsini, penalty


class ParamMapper:
    """vel_fit.ParamMapper (vel_fit.py:119-202) for a batch: parameter vectors
    [J, n] ordered (vel, [vsini], free stellar parameters in specParams order)
    <-> vel [J], vsini [J] | None, params [J, ndim], penalty [J].
    paramDict0 values are [S] tensors; `idx` selects the spectra of the J rows."""

    def __init__(self, specParams, paramDict0, fixParam, vsiniMapper,
                 fitVsini=True):
        self.specParams = specParams
        self.paramDict0 = paramDict0
        self.fixParam = fixParam
        self.vsiniMapper = vsiniMapper
        self.fitVsini = fitVsini

    def forward(self, p0, idx=None):
        if idx is None and not isinstance(p0, torch.Tensor):
            return self._forward_one(p0)
        ret = {}
        k = 0
        ret['vel'] = p0[:, k]
        k += 1
        penalty = torch.zeros_like(ret['vel'])
        if self.fitVsini:
            vsini, pen = self.vsiniMapper.to_vsini(p0[:, k])
            k += 1
            penalty = penalty + pen
            ret['vsini'] = vsini
        elif 'vsini' in self.fixParam:
            ret['vsini'] = self.paramDict0['vsini'][idx]
        else:
            ret['vsini'] = None
        cols = []
        for x in self.specParams:
            if x in self.fixParam:
                cols.append(self.paramDict0[x][idx])
            else:
                cols.append(p0[:, k])
                k += 1
        assert k == p0.shape[1]
        ret['params'] = torch.stack(cols, dim=1)
        ret['penalty'] = penalty
        return ret

    def _forward_one(self, p0):
        """the reference's own form (vel_fit.py:156-202): ONE parameter vector ->
        dict(vel, vsini, rot_params, params [list], penalty) of plain numbers;
        paramDict0 holds numbers"""
        ret = {}
        p0rev = list(p0)[::-1]
        penalty = 0
        ret['vel'] = p0rev.pop()
        if self.fitVsini:
            vsini, penalty_vsini = self.vsiniMapper.to_vsini(p0rev.pop())
            penalty += penalty_vsini
            ret['vsini'] = vsini
        elif 'vsini' in self.fixParam:
            ret['vsini'] = self.paramDict0['vsini']
        else:
            ret['vsini'] = None
        ret['rot_params'] = None if ret['vsini'] is None else (ret['vsini'], )
        ret['params'] = []
        for x in self.specParams:
            if x in self.fixParam:
                ret['params'].append(self.paramDict0[x])
            else:
                ret['params'].append(p0rev.pop())
        assert len(p0rev) == 0
        ret['penalty'] = penalty
        return ret

    def get_fitted_params(self):
        ret = ['vel']
        if self.fitVsini:
            ret.append('vsini')
        for x in self.specParams:
            if x not in self.fixParam:
                ret.append(x)
        return ret


def chisq_func0(pdict, args, outside_penalty=True):
    """vel_fit.chisq_func0 (vel_fit.py:210-230): chi-square + priors of ONE spectrum
    at the parameter dictionary `pdict` (ParamMapper.forward of one vector); `args` as
    the reference builds it in process (specdata, paramMapper, resolParams, options,
    config, priors).  For user code that drives its own sampler over the objective of
    `process`; process itself evaluates S such objectives per kernel launch."""
    chisq = 0
    if args.get('priors') is not None:
        priors = args['priors']
        for i, k in enumerate(args['paramMapper'].specParams):
            if k in priors:
                chisq += ((priors[k][0] - pdict['params'][i]) / priors[k][1])**2
    chisq += spec_fit.get_chisq(args['specdata'], pdict['vel'],
                                tuple(pdict['params']), pdict['rot_params'],
                                args.get('resolParams'), options=args['options'],
                                config=args['config'],
                                outside_penalty=outside_penalty)
    return chisq


def chisq_func(p, args):
    """vel_fit.chisq_func (vel_fit.py:233-257): the function process minimises --
    1e30 outside [min_vel, max_vel] or at a non-finite parameter, else chisq_func0 +
    the vsini penalty"""
    pdict = args['paramMapper'].forward(p)
    if (pdict['vel'] > args['max_vel'] or pdict['vel'] < args['min_vel']
            or (~np.isfinite(np.asarray(pdict['params'], dtype=float))).any()):
        return 1e30
    return chisq_func0(pdict, args) + pdict['penalty']


def hess_func(p, pdict, args):
    """vel_fit.hess_func (vel_fit.py:260-269): 0.5 chi-square in the stellar parameters
    `p` at the rest of `pdict`"""
    pdict['params'][:] = p[:]
    return 0.5 * chisq_func0(pdict, args)


def get_hess_inv(param_names):
    """vel_fit.get_hess_inv (vel_fit.py:442-461)"""
    diag = np.zeros(len(param_names)) + 0.1**2
    names = np.asarray(param_names)
    diag[np.nonzero(names == 'teff')[0][0]] = 50**2
    vi = np.nonzero(names == 'vsini')[0]
    if len(vi) == 1:
        diag[vi] = 5**2
    diag[0] = 1**2
    return np.diag(diag)


def _uncertainties_from_hessians(H):
    """_uncertainties_from_hessian for a stack [S, n, n] in one LAPACK call (the
    per-matrix Python loop was 0.6 s per 10 000 spectra): np.linalg.inv factors
    every matrix of a stack on its own, so each inverse is the one-matrix
    call's, bit for bit; a singular or non-finite matrix takes the
    diagonal-only branch of the reference.  Returns (diag_err [S, n], covar
    [S, n, n], bad [S])."""
    H = np.asarray(H, dtype=np.float64)
    S, n = H.shape[0], H.shape[1]
    dh = np.diagonal(H, axis1=1, axis2=2)
    with np.errstate(all='ignore'):
        inv_d = 1. / (dh + (dh == 0))
    inv_d = np.where(dh == 0, np.inf, inv_d)
    bad = np.zeros(S, dtype=bool)
    Hinv = np.empty_like(H)
    ok = np.isfinite(H).all(axis=(1, 2))
    if ok.any():
        try:
            Hinv[ok] = np.linalg.inv(H[ok])
        except np.linalg.LinAlgError:    # some matrix is exactly singular
            for i in np.nonzero(ok)[0]:
                try:
                    Hinv[i] = np.linalg.inv(H[i])
                except np.linalg.LinAlgError:
                    ok[i] = False
    for i in np.nonzero(~ok)[0]:
        bad[i] = True
        Hinv[i] = np.diag(inv_d[i])
    e0 = np.array(np.diagonal(Hinv, axis1=1, axis2=2))
    e1 = inv_d
    bad0, bad1 = e0 < 0, e1 < 0
    bad |= bad0.any(axis=1)
    sub1, sub2 = bad0 & ~bad1, bad0 & bad1
    e0[sub1] = e1[sub1]
    e0[sub2] = 0
    with np.errstate(all='ignore'):
        err = np.sqrt(e0)
    err[sub2] = np.nan
    bad |= (~np.isfinite(err)).any(axis=1)
    return err, Hinv, bad


def _uncertainties_from_hessian(hessian):
    """vel_fit._uncertainties_from_hessian (vel_fit.py:464-502), one matrix"""
    diag_hessian = np.diag(hessian)
    with np.errstate(all='ignore'):
        inv_diag_hessian = 1. / (diag_hessian + (diag_hessian == 0))
    inv_diag_hessian[diag_hessian == 0] = np.inf
    bad_hessian = False
    try:
        if not np.isfinite(hessian).all():
            raise ValueError('non finite Hessian')  # scipy.linalg.inv check_finite
        hessian_inv = np.linalg.inv(hessian)
    except (np.linalg.LinAlgError, ValueError):
        bad_hessian = True
        hessian_inv = np.diag(inv_diag_hessian)
    diag_err0 = np.array(np.diag(hessian_inv))
    diag_err1 = inv_diag_hessian
    bad_err0 = diag_err0 < 0
    bad_err1 = diag_err1 < 0
    if bad_err0.any():
        bad_hessian = True
    sub1 = bad_err0 & (~bad_err1)
    sub2 = bad_err0 & bad_err1
    diag_err0[sub1] = diag_err1[sub1]
    diag_err0[sub2] = 0
    with np.errstate(all='ignore'):
        diag_err = np.sqrt(diag_err0)
    diag_err[sub2] = np.nan
    if (~np.isfinite(diag_err)).sum() != 0:
        bad_hessian = True
    return diag_err, hessian_inv, bad_hessian


# iteration cap of a Nelder-Mead run (vel_fit.py:630, 644: maxiter=10000); a
# spectrum that hits it is restarted once from its final simplex
NM_MAXITER = 10000
_SIMPLEX_STD = {'logg': 0.5, 'teff': 300, 'feh': 0.5, 'alpha': 0.25}
HESS_BASE_STEP = {'vsini': 1 / 100, 'logg': 0.1 / 100, 'feh': 0.1 / 100,
                  'alpha': .01 / 100, 'teff': 1 / 100, 'vrad': 1 / 100}


def _get_simplex_start(best_vel, fixParam, specParamNames, paramDict0,
                       vsiniMapper, fitVsini):
    """vel_fit._get_simplex_start (vel_fit.py:272-312) for [S] starting points:
    the same deterministic RandomState(43434) displacement pattern for every
    spectrum."""
    cols = [best_vel]
    std_vec = [5]
    if fitVsini:
        cols.append(vsiniMapper.to_internal(paramDict0['vsini']))
        std_vec.append(3)
    for x in specParamNames:
        if x not in fixParam:
            cols.append(paramDict0[x])
            std_vec.append(_SIMPLEX_STD.get(x) or 0.5)
    curval = torch.stack(cols, dim=1)
    ndim = curval.shape[1]
    R = np.random.RandomState(43434)
    disp = np.array(std_vec)[None, :] * R.normal(size=(ndim, ndim))
    simp = curval[:, None, :].repeat(1, ndim + 1, 1)
    simp[:, 1:, :] = curval[:, None, :] + torch.as_tensor(disp).to(curval.device)
    return curval, simp


class _Objective:
    """chisq_func / chisq_func0 (vel_fit.py:205-254) for J jobs at a time."""

    def __init__(self, batch, mapper, config, options, priors,
                 resolParams=None):
        self.resolParams = resolParams
        self.batch, self.mapper = batch, mapper
        self.config, self.options, self.priors = config, options, priors
        self.min_vel, self.max_vel = config['min_vel'], config['max_vel']
        self.status = torch.zeros(batch.S, dtype=torch.int32,
                                  device=batch.device)
        self.nfev = 0

    def chisq0(self, idx, vel, params, vsini):
        chisq = torch.zeros_like(vel)
        if self.priors is not None:
            for i, k in enumerate(self.mapper.specParams):
                if k in self.priors:
                    m, sg = self.priors[k]
                    m = m[idx] if isinstance(m, torch.Tensor) else m
                    sg = sg[idx] if isinstance(sg, torch.Tensor) else sg
                    chisq = chisq + ((m - params[:, i]) / sg)**2
        c, st = spec_fit.chisq_jobs(self.batch, idx, vel, params, vsini,
                                    self.options, self.config,
                                    resol_params=self.resolParams)
        self.status[idx] |= st
        self.nfev += idx.numel()
        return chisq + c

    def __call__(self, idx, p):
        pd = self.mapper.forward(p, idx)
        vel, params = pd['vel'], pd['params']
        bad = (vel > self.max_vel) | (vel < self.min_vel) | \
            (~torch.isfinite(params)).any(dim=1)
        # rows that the reference answers with 1e30 without evaluating are
        # evaluated at a harmless point and overwritten
        velc = torch.where(bad, torch.zeros_like(vel), vel)
        parc = torch.where(bad[:, None], self.safe_params[idx], params)
        ret = self.chisq0(idx, velc, parc, pd['vsini']) + pd['penalty']
        return torch.where(bad, torch.full_like(ret, 1e30), ret)


def _as_param_tensors(paramDict0, S, dev):
    out = {}
    for k, v in paramDict0.items():
        if isinstance(v, torch.Tensor):
            t = v.to(dev, torch.float64).reshape(-1)
        else:
            t = torch.as_tensor(np.asarray(v, dtype=np.float64)).to(dev
                                                                     ).reshape(-1)
        out[k] = t.expand(S).contiguous() if t.numel() == 1 else t.contiguous()
        assert out[k].shape[0] == S
    return out


def _hessian_stage(obj, names, vel, params, vsini):
    """vel_fit.py:699-725 for S spectra: the Hessian of 0.5 * chisq_func0 in the
    stellar parameters at (vel, params, vsini), as numdifftools computes it
    (numdiff.py), and vel_fit._uncertainties_from_hessian per spectrum.
    First ndf.Hessian(step=MinStepGenerator(base_step)) = the central rule at
    one step; rows flagged bad are redone with the default step generator
    (15 steps, Richardson + Wynn extrapolation).  Returns numpy arrays
    (param_err [S, n], param_covar [S, n, n], bad_hessian [S])."""
    dev = params.device
    S = params.shape[0]

    def hess_func(idx, p):   # vel_fit.hess_func (vel_fit.py:257-269)
        return 0.5 * obj.chisq0(idx, vel[idx], p.contiguous(),
                                None if vsini is None else vsini[idx])

    base = torch.as_tensor([HESS_BASE_STEP[_] for _ in names],
                           dtype=torch.float64, device=dev)
    bad_hessian = np.zeros(S, dtype=bool)
    diag_err = np.zeros((S, len(names)))
    covar = np.zeros((S, len(names), len(names)))
    todo = np.arange(S)
    for attempt in range(2):
        if len(todo) == 0:
            break
        tt = torch.as_tensor(todo).to(dev)
        xx = params[tt]
        fn = lambda i, p: hess_func(tt[i], p)   # noqa: E731
        if attempt == 0:
            H = numdiff.hessian_central(fn, xx, numdiff.first_try_step(base, xx)
                                        ).cpu().numpy()
        else:
            H = numdiff.hessian_retry(fn, xx)
        diag_err[todo], covar[todo], bad_hessian[todo] = \
            _uncertainties_from_hessians(H)
        todo = todo[bad_hessian[todo]]
    return diag_err, covar, bad_hessian


def param_uncertainties(specdata, vel, atm_params, vsini=None, options=None,
                        config=None, resolParams=None, priors=None):
    """The uncertainty stage of `process` on its own (vel_fit.py:699-725): the
    finite-difference Hessian of 0.5 chi^2 in the stellar parameters at the given
    point.  One spectrum (list of SpecData; atm_params a dict or sequence in
    the interpolator's parameter order): dict(param_err {name: sigma},
    param_covar, bad_hessian); a SpecBatch with [S] / [S, n] tensors: arrays
    with a leading S axis."""
    batch, is_batch = as_batch(specdata)
    options = options or {}
    dev = batch.device
    S = batch.S
    names = list(spec_inter.getSpecParams(batch.names[0], config))
    if isinstance(atm_params, dict):
        atm_params = [atm_params[k] for k in names]
    pt = torch.as_tensor(np.asarray(
        atm_params.cpu() if isinstance(atm_params, torch.Tensor) else atm_params,
        dtype=np.float64)).to(dev).reshape(-1, len(names))
    pt = pt.expand(S, len(names)).contiguous()
    pd = _as_param_tensors(dict(vel=vel), S, dev)
    vs = None
    if vsini is not None:
        vs = _as_param_tensors(dict(vsini=vsini), S, dev)['vsini']
    mapper = types.SimpleNamespace(specParams=names)
    obj = _Objective(batch, mapper, config, options, priors, resolParams)
    err, covar, bad = _hessian_stage(obj, names, pd['vel'], pt, vs)
    if is_batch:
        return dict(param_err={k: err[:, i] for i, k in enumerate(names)},
                    param_covar=covar, bad_hessian=bad, status=obj.status)
    return dict(param_err=dict(zip(names, err[0])), param_covar=covar[0],
                bad_hessian=bool(bad[0]))


# A SpecBatch of at least PROCESS_SPLIT_MIN spectra is fitted as two interleaved
# halves by two host threads on two HIP streams: the optimiser's rounds run in C
# (rvs_nm_run, no interpreter lock), so while one half is in the latency-bound
# tail of its rounds the other's kernels fill the CUs.  Results are those of the
# unsplit run, bit for bit (no spectrum sees another;
# test_process_two_halves_equal_one_batch).  1 switches it off.
PROCESS_STREAMS = int(os.environ.get('RVS_PROCESS_STREAMS', '2'))
# the second minimiser's rounds on the device (rvs_bfgs_run) wherever the library
# launches the objective itself; False: the host machines around the Python objective
# (what Delaunay evaluators and resolution matrices take anyway) -- the two run the
# same state machine (test_process_bfgs_device_equals_host)
BFGS_ON_DEVICE = os.environ.get('RVS_BFGS_ON_DEVICE', '1') != '0'
# (64: the stellar targets of one DESI petal are 100-200 spectra -- split, 100 spectra
# 832 -> 879 per second, 200: 1207 -> 1272; it was 256 until the end of round 5)
PROCESS_SPLIT_MIN = int(os.environ.get('RVS_PROCESS_SPLIT_MIN', '64'))


_tls = threading.local()


@contextlib.contextmanager
def single_stream():
    """process() calls of this thread inside the block are not split: for
    callers whose own host threads already keep the GPU and the interpreter busy
    (desi_fit.proc_many, which conditions the next group of files meanwhile --
    there the split measured 5 % slower)."""
    old = getattr(_tls, 'single', False)
    _tls.single = True
    try:
        yield
    finally:
        _tls.single = old


def _merge_parts(parts, idxs, S):
    """results of process() on the spectra idxs[k] -> one result over S"""
    first = parts[0]
    n0 = len(idxs[0])

    def merge(vals):
        v = vals[0]
        if isinstance(v, dict):
            return {k: merge([x[k] for x in vals]) for k in v}
        if isinstance(v, (list, tuple)):
            return [merge([x[i] for x in vals]) for i in range(len(v))]
        if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == n0:
            if v.dim() == 2 and len({x.shape[1] for x in vals}) > 1:
                # per-pixel rows of a grid set: each part is as wide as the longest
                # grid IT holds (ArmData.subset); zeros behind a spectrum's pixels
                out = torch.zeros((S, max(x.shape[1] for x in vals)),
                                  dtype=v.dtype, device=v.device)
                for x, ix in zip(vals, idxs):
                    out[ix, :x.shape[1]] = x
                return out
            out = torch.empty((S, ) + tuple(v.shape[1:]), dtype=v.dtype,
                              device=v.device)
            for x, ix in zip(vals, idxs):
                out[ix] = x
            return out
        if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == n0:
            out = np.empty((S, ) + v.shape[1:], dtype=v.dtype)
            for x, ix in zip(vals, idxs):
                out[ix.cpu().numpy()] = x
            return out
        return v
    out = {k: merge([p[k] for p in parts]) for k in first}
    out['nm_rounds'] = max(p['nm_rounds'] for p in parts)
    out['objective_evals'] = sum(p['objective_evals'] for p in parts)
    out['nm_launched_rows'] = sum(p.get('nm_launched_rows', 0) for p in parts)
    if 'bfgs' in first:
        out['bfgs']['rounds'] = max(p['bfgs']['rounds'] for p in parts)
    return out


_SPLIT_STREAMS = {}


@contextlib.contextmanager
def stream_lane(lane):
    """process() calls of this thread inside the block run their halves on the
    streams of `lane` (desi_fit.proc_many with two fit threads: a pair of streams
    per thread, so that two groups' rounds do not queue behind one another)"""
    old = getattr(_tls, 'lane', 0)
    _tls.lane = int(lane)
    try:
        yield
    finally:
        _tls.lane = old


def _split_stream(dev, k, lane=0):
    # kept: the caching allocator's pools are per stream
    key = (dev.index, k, lane)
    if key not in _SPLIT_STREAMS:
        _SPLIT_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SPLIT_STREAMS[key]


def _process_split(batch, paramDict0, kwargs):
    S, dev = batch.S, batch.device
    nparts = max(2, int(PROCESS_STREAMS))
    idxs = [torch.arange(k, S, nparts, device=dev) for k in range(nparts)]
    pd = _as_param_tensors(paramDict0, S, dev)
    pri = kwargs.get('priors')
    # template libraries are loaded (uploaded) here, before either thread
    spec_inter.get_libs(batch.names, kwargs['config'])
    torch.cuda.current_stream().synchronize()
    parts, errs = [None] * nparts, [None] * nparts
    lane = getattr(_tls, 'lane', 0)

    def run(k):
        try:
            st = _split_stream(dev, k, lane)
            with torch.cuda.stream(st):
                sub = batch.subset(idxs[k])
                pdk = {n: v[idxs[k]].contiguous() for n, v in pd.items()}
                kw = dict(kwargs)
                if pri:
                    # per-spectrum priors are tensors (see _Objective.chisq0)
                    kw['priors'] = {
                        n: tuple(x[idxs[k]].contiguous()
                                 if isinstance(x, torch.Tensor) and x.dim()
                                 else x for x in mv) for n, mv in pri.items()}
                parts[k] = _process_one(sub, pdk, **kw)
            st.synchronize()
        except BaseException as e:  # noqa: BLE001 -- re-raised by the caller
            errs[k] = e
    th = [threading.Thread(target=run, args=(k, )) for k in range(nparts)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in errs:
        if e is not None:
            raise e
    return _merge_parts(parts, idxs, S)


def _rounds_run_in_c(batch, config, resolParams, options=None):
    """the optimiser's rounds of this batch run inside the library (rvs_nm_run:
    regular-grid and MLP libraries, no resolution matrix); only then do two host
    threads help -- rounds driven from Python (Delaunay evaluators, resolution
    matrices) share the interpreter lock (round 3, NN rounds still in Python:
    432 against 586 spectra/s split in two)"""
    libs = spec_inter.get_libs(batch.names, config)
    npoly = (options or {}).get('npoly') or 5
    rs = spec_fit._resols(batch, resolParams)
    if engine.can_fuse_objective(batch, libs, rs, npoly=npoly):
        return True
    # MLP libraries on every arm: rvs_nm_run drives rvs_template_nn +
    # rvs_objective_from_template itself
    def native(kind):
        return all(libs[a.name].kind == kind for a in batch.arms)
    return (native('nn') or (native('triangulation') and all(
        libs[a.name]._tri_bk is not None for a in batch.arms))) and \
        engine.can_fuse_objective(batch, libs, rs, npoly=npoly,
                                  from_template=True)


def process(specdata, paramDict0, fixParam=None, options=None, config=None,
            resolParams=None, priors=None, timers=None):
    """vel_fit.process (see _process_one); a large SpecBatch is fitted as two
    concurrent halves (PROCESS_STREAMS)."""
    if config is None:
        raise RuntimeError('Config must be provided')
    if (PROCESS_STREAMS >= 2 and isinstance(specdata, SpecBatch)
            and specdata.S >= PROCESS_SPLIT_MIN and timers is None
            and not getattr(_tls, 'single', False)
            and _rounds_run_in_c(specdata, config, resolParams, options)):
        return _process_split(specdata, paramDict0, dict(
            fixParam=fixParam, options=options, config=config,
            resolParams=resolParams, priors=priors))
    return _process_one(specdata, paramDict0, fixParam=fixParam, options=options,
                        config=config, resolParams=resolParams, priors=priors,
                        timers=timers)


def _process_one(specdata, paramDict0, fixParam=None, options=None, config=None,
                 resolParams=None, priors=None, timers=None):
    """vel_fit.process (vel_fit.py:505-737): velocity grid at the starting
    parameters -> Nelder-Mead (deterministic start simplex, fatol 1e-3, xatol
    1e-2, up to two runs) -> [BFGS] -> velocity refinement -> full output ->
    finite-difference Hessian -> parameter uncertainties.

    One spectrum (list of SpecData) returns the reference's dict.  A SpecBatch
    (paramDict0 values scalars or [S] arrays) returns [S]-leading device
    tensors; the S optimisers advance in lock-step (optimizer.DeviceNelderMead), every
    objective evaluation is one batched template build + chi^2 launch set.

    config['second_minimizer'] (default True in utils.read_config, as in the
    reference) adds scipy's BFGS (csrc/bfgs_host.cpp; host state per spectrum,
    batched objective).  The Hessian is numdifftools' (vel_fit.py:713-716)
    restated in numdiff.py: MinStepGenerator(base_step) -> one central step;
    the step=None retry of rows with a bad Hessian -> 15 steps, Richardson +
    Wynn extrapolation; every displacement pattern is one batched objective
    call (param_err pinned to 1e-3: test_param_uncertainties_at_reference_optimum)."""
    if config is None:
        raise RuntimeError('Config must be provided')
    options = options or {}
    batch, is_batch = as_batch(specdata)
    S, dev = batch.S, batch.device
    min_vel, max_vel = config['min_vel'], config['max_vel']
    vel_step0 = config['vel_step0']
    max_vsini = config['max_vsini']
    min_vel_step = config['min_vel_step']
    fixParam = list(fixParam) if fixParam is not None else []
    names = spec_inter.getSpecParams(batch.names[0], config)
    pd0 = _as_param_tensors(paramDict0, S, dev)
    curparam = torch.stack([pd0[_] for _ in names], dim=1)
    vsiniMapper = None
    if 'vsini' not in pd0:
        vsini0, fitVsini = None, False
    else:
        vsini0 = pd0['vsini']
        fitVsini = 'vsini' not in fixParam
        if fitVsini:
            vsiniMapper = VSiniMapper(max_vsini)
    tm = timers if timers is not None else {}

    def _tick(k, t0):
        # (stage clocks only when somebody asked for them: the device-wide
        # synchronisation they need would make the two halves of a split batch wait
        # for one another at every stage boundary)
        if timers is None:
            return
        if str(dev).startswith('cuda'):
            torch.cuda.synchronize()
        tm[k] = tm.get(k, 0.) + time.time() - t0

    # vel_fit.py:596-602
    t0 = time.time()
    vg = torch.as_tensor(np.arange(min_vel, max_vel, vel_step0,
                                   dtype=np.float64)).to(dev)
    chisq, st0, _ = spec_fit.chisq_grid_jobs(batch, vg, curparam[:, None, :],
                                             vsini0, options, config,
                                             resol_params=resolParams)
    res, _, mst = engine.grid_moments(chisq.reshape(S, -1), vg, Np=1)
    best_vel = res[:, 1].contiguous()
    _tick('grid0', t0)

    t0 = time.time()
    curval, simplex = _get_simplex_start(best_vel, fixParam, names, pd0,
                                         vsiniMapper, fitVsini)
    mapper = ParamMapper(names, pd0, fixParam, vsiniMapper, fitVsini=fitVsini)
    obj = _Objective(batch, mapper, config, options, priors, resolParams)
    obj.safe_params = curparam
    stats = {}
    # vel_fit.py:624-649: a second run restarts from the final simplex
    libs = spec_inter.get_libs(batch.names, config)
    from . import optimizer

    # The spectra that leave the simplex stage first do not wait for the slowest
    # simplex (_early_split): their BFGS polish, refinement and Hessian run on a second
    # stream under the stage's latency-bound last rounds.
    early = (EARLY_SPLIT and is_batch and timers is None and S >= EARLY_SPLIT_MIN
             and optimizer.NATIVE_ROUNDS)
    pobj = optimizer.ProcessObjective(
        batch, libs, names, pd0, fixParam, fitVsini, config, options, priors,
        curparam, resols=spec_fit._resols(batch, resolParams))
    nmdev = optimizer.DeviceNelderMead(S, simplex.shape[2], dev)
    early = early and (pobj.fused or pobj.nn_native or pobj.tri_native)
    kw_nm = dict(stop_below=max(1, int(EARLY_SPLIT_FRAC * S))) if early else {}
    nm = nmdev.minimize(pobj, simplex, fatol=1e-3, xatol=1e-2, maxiter=NM_MAXITER,
                        stats=stats, **kw_nm)
    ctx = dict(batch=batch, pd0=pd0, priors=priors, curparam=curparam, names=names,
               fixParam=fixParam, fitVsini=fitVsini, vsiniMapper=vsiniMapper,
               config=config, options=options, resolParams=resolParams, libs=libs,
               st0=st0.reshape(S, -1)[:, 0], tick=_tick, is_batch=is_batch)
    side = None
    if nm.get('paused'):
        fin = torch.nonzero(nm['finished']).reshape(-1)
        rest = torch.nonzero(~nm['finished']).reshape(-1)
        if fin.numel() >= EARLY_SPLIT_MIN // 4 and rest.numel() > 0:
            # the finished spectra's later stages, on a stream of their own
            st_main = torch.cuda.current_stream()
            key = (dev.index, st_main.cuda_stream)
            if key not in _EARLY_STREAMS:
                _EARLY_STREAMS[key] = torch.cuda.Stream(device=dev)
            st_side = _EARLY_STREAMS[key]
            ev = torch.cuda.Event()
            ev.record(st_main)
            xs, ns, fs_ = nm['x'][fin].clone(), nm['nit'][fin].clone(), \
                nm['nfev'][fin].clone()
            stat_fin = pobj.status[fin].clone()
            box = {}

            def run_side():
                try:
                    with torch.cuda.stream(st_side):
                        st_side.wait_event(ev)
                        box['ret'] = _post_nm(
                            ctx, fin, xs, torch.ones_like(fin, dtype=torch.bool), ns,
                            fs_, stat_fin, None, dict(rounds=0, evals=0, slots=0))
                    st_side.synchronize()
                except BaseException as e:  # noqa: BLE001 -- re-raised below
                    box['err'] = e
            side = (threading.Thread(target=run_side), fin, rest, box)
            side[0].start()
            EARLY_SPLITS.append((int(fin.numel()), int(rest.numel())))
        nm = nmdev.resume(pobj, stats=stats)
    obj.status |= pobj.status
    obj.nfev += pobj.jobs
    slots = getattr(pobj, 'slots', 0)
    success = nm['success']
    x, nit, nfev = nm['x'], nm['nit'], nm['nfev']
    redo = torch.nonzero(~success).reshape(-1)
    if redo.numel():
        # vel_fit.py:624-649: a second run restarts from the final simplex; the
        # unconverged spectra form a batch of their own on the same kernels
        pri2 = _subset_priors(priors, redo)
        sub = batch.subset(redo)
        pobj2 = optimizer.ProcessObjective(
            sub, libs, names, {k_: v[redo].contiguous() for k_, v in pd0.items()},
            fixParam, fitVsini, config, options, pri2, curparam[redo].contiguous(),
            resols=spec_fit._resols(sub, resolParams))
        nm2 = optimizer.DeviceNelderMead(sub.S, simplex.shape[2], dev).minimize(
            pobj2, nm['final_simplex'][0][redo].contiguous(), fatol=1e-3,
            xatol=1e-2, maxiter=NM_MAXITER, stats=stats)
        obj.status[redo] |= pobj2.status
        obj.nfev += pobj2.jobs
        slots += getattr(pobj2, 'slots', 0)
        x[redo] = nm2['x']
        success[redo] = nm2['success']
        nit[redo] += nm2['nit']
        nfev[redo] += nm2['nfev']
    _tick('neldermead', t0)
    nmstats = dict(rounds=stats.get('rounds', 0), evals=obj.nfev, slots=slots)
    if side is None:
        return _post_nm(ctx, None, x, success, nit, nfev, obj.status, pobj, nmstats)
    th, fin, rest, box = side
    r_rest = _post_nm(ctx, rest, x[rest].contiguous(), success[rest], nit[rest],
                      nfev[rest], obj.status[rest], None, nmstats)
    th.join()
    if 'err' in box:
        raise box['err']
    torch.cuda.current_stream().synchronize()
    return _merge_parts([box['ret'], r_rest], [fin, rest], S)


# An option, OFF by default (measured slower, below): spectra that finish the simplex
# stage early go on to their BFGS polish, refinement and Hessian while the stragglers'
# last rounds -- a chain of latencies that leaves the chip mostly idle -- are still
# running: rvs_nm_run returns at its first look that finds at most EARLY_SPLIT_FRAC of
# the simplices running, the converged spectra's later stages start on a second stream,
# and the rounds resume.  No spectrum sees another: the results are those of the
# unsplit run bit for bit (test_process_early_split_equals_unsplit).  What it costs is
# a second latency-bound BFGS tail (the stragglers' own ~200 rounds, behind
# everything) and the sub-batches' set-up: --process 2000 2060-2110 against 2190-2200
# spectra/s for fractions of 0.02-0.25, 500: 1340-1410 against 1480-1505, 10 000 equal
# (tools/perf/proc_ab.sh, one job, alternating).
EARLY_SPLIT = os.environ.get('RVS_EARLY_SPLIT', '0') != '0'
EARLY_SPLIT_FRAC = float(os.environ.get('RVS_EARLY_SPLIT_FRAC', '0.25'))
EARLY_SPLIT_MIN = int(os.environ.get('RVS_EARLY_SPLIT_MIN', '128'))
_EARLY_STREAMS = {}
EARLY_SPLITS = []   # (finished, still running) of every split made (tests, tools)


def _subset_priors(priors, idx):
    if not priors:
        return priors
    return {n_: tuple(v[idx].contiguous()
                      if isinstance(v, torch.Tensor) and v.dim() else v for v in mv)
            for n_, mv in priors.items()}


def _post_nm(ctx, idx, x, success, nit, nfev, status_nm, pobj, nmstats):
    """vel_fit.py:653-737 behind the simplex stage, for the spectra `idx` of the
    batch (None: all of them, on the simplex stage's own objective `pobj`): BFGS
    polish, velocity refinement, full output, Hessian, the result dict."""
    from . import optimizer
    batch, pd0, priors, curparam = ctx['batch'], ctx['pd0'], ctx['priors'], \
        ctx['curparam']
    names, fixParam, fitVsini = ctx['names'], ctx['fixParam'], ctx['fitVsini']
    config, options, resolParams = ctx['config'], ctx['options'], ctx['resolParams']
    _tick, is_batch, st0 = ctx['tick'], ctx['is_batch'], ctx['st0']
    if idx is not None:
        batch = batch.subset(idx)
        pd0 = {k_: v[idx].contiguous() for k_, v in pd0.items()}
        priors = _subset_priors(priors, idx)
        curparam = curparam[idx].contiguous()
        st0 = st0[idx]
    S, dev = batch.S, batch.device
    mapper = ParamMapper(names, pd0, fixParam, ctx['vsiniMapper'], fitVsini=fitVsini)
    obj = _Objective(batch, mapper, config, options, priors, resolParams)
    obj.safe_params = curparam
    obj.status |= status_nm
    slots = nmstats['slots']
    allidx = torch.arange(S, device=dev)
    # vel_fit.py:653-658: optional BFGS polish from the simplex optimum
    second_run = False
    bfgs_info = None
    if config.get('second_minimizer'):
        from . import bfgs
        t0 = time.time()
        hess_inv0 = get_hess_inv(mapper.get_fitted_params())
        if pobj is None and BFGS_ON_DEVICE and optimizer.NATIVE_ROUNDS:
            pobj = optimizer.ProcessObjective(
                batch, ctx['libs'], names, pd0, fixParam, fitVsini, config, options,
                priors, curparam, resols=spec_fit._resols(batch, resolParams))
        if BFGS_ON_DEVICE and optimizer.NATIVE_ROUNDS and pobj is not None and (
                pobj.fused or pobj.nn_native or pobj.tri_native):
            # the rounds inside the library (rvs_bfgs_run), on the objective the
            # simplex stage ran on
            jobs_before = pobj.jobs
            slots_before = getattr(pobj, 'slots', 0)
            br = bfgs.minimize_lockstep_device(pobj, x, hess_inv0=hess_inv0)
            obj.status |= pobj.status
            obj.nfev += pobj.jobs - jobs_before
            slots += getattr(pobj, 'slots', 0) - slots_before
            x = br['x']
            bfgs_info = dict(nit=br['nit'].cpu().numpy(),
                             nfev=br['nfev'].cpu().numpy(),
                             status=br['status'].cpu().numpy(),
                             rounds=br['rounds'], device=True)
        else:
            def rows(idx_np, X_np):
                it = torch.as_tensor(idx_np).to(dev)
                return obj(it, torch.as_tensor(X_np).to(dev)).cpu().numpy()

            br = bfgs.minimize_lockstep_native(rows, x.cpu().numpy(),
                                               hess_inv0=hess_inv0,
                                               max_rows=max(S, 1024))
            x = torch.as_tensor(br['x']).to(dev)
            bfgs_info = dict(nit=br['nit'], nfev=br['nfev'], status=br['status'],
                             rounds=br['rounds'], device=False)
        second_run = True
        _tick('bfgs', t0)
    best = mapper.forward(x, allidx)
    nm_vel = best['vel'].contiguous()
    bparams = best['params'].contiguous()
    bvsini = best['vsini']

    # vel_fit.py:672-682
    t0 = time.time()
    vel_in = nm_vel.cpu().numpy()
    r = _minimum_sampler_batch(batch, vel_in, bparams, bvsini, config, options,
                               resolParams=resolParams)
    best_vel = torch.as_tensor(r['best_vel']).to(dev)
    _tick('vel_refine', t0)

    # vel_fit.py:689-696
    t0 = time.time()
    outp = spec_fit.get_chisq(batch, best_vel, bparams,
                              None if bvsini is None else bvsini, resolParams,
                              options=options, config=config, full_output=True)
    _tick('full_output', t0)

    # vel_fit.py:699-725: Hessian of 0.5*chisq_func0 in ALL stellar parameters
    # at the optimiser's velocity (best_param is not updated by the refinement)
    t0 = time.time()

    diag_err, covar, bad_hessian = _hessian_stage(obj, names, nm_vel, bparams,
                                                  bvsini)
    _tick('hessian', t0)

    ret = {}
    if is_batch:
        ret['param'] = {k: bparams[:, i] for i, k in enumerate(names)}
        if fitVsini:
            ret['vsini'] = bvsini
        ret['vel'] = best_vel
        ret['vel_err'] = torch.as_tensor(r['vel_err']).to(dev)
        ret['vel_skewness'] = torch.as_tensor(r['skewness']).to(dev)
        ret['vel_kurtosis'] = torch.as_tensor(r['kurtosis']).to(dev)
        ret['param_err'] = {k: diag_err[:, i] for i, k in enumerate(names)}
        ret['param_covar'] = covar
        ret['minimize_success'] = success
        ret['bad_hessian'] = bad_hessian
        ret['yfit'] = outp['models']
        ret['raw_models'] = outp['raw_models']
        ret['chisq'] = outp['chisq']
        ret['logl'] = outp['logl']
        ret['chisq_array'] = outp['chisq_array']
        ret['npix_array'] = outp['npix_array']
        ret['status'] = obj.status | outp['status'] | st0
    else:
        if int(obj.status[0].item()) & _lib_nonfinite():
            raise RuntimeError('non-finite likelihood during the optimisation')
        bp = bparams[0].cpu().numpy()
        ret['param'] = dict(zip(names, [float(_) for _ in bp]))
        if fitVsini:
            ret['vsini'] = float(bvsini[0].item())
        ret['vel'] = float(r['best_vel'][0])
        ret['vel_err'] = float(r['vel_err'][0])
        ret['vel_skewness'] = float(r['skewness'][0])
        ret['vel_kurtosis'] = float(r['kurtosis'][0])
        ret['param_err'] = dict(zip(names, diag_err[0]))
        ret['param_covar'] = covar[0]
        ret['minimize_success'] = bool(success[0].item())
        ret['bad_hessian'] = bool(bad_hessian[0])
        ret['yfit'] = [m[0].cpu().numpy() for m in outp['models']]
        ret['raw_models'] = [m[0].cpu().numpy() for m in outp['raw_models']]
        ret['chisq'] = float(outp['chisq'][0].item())
        ret['logl'] = -0.5 * ret['chisq']
        ret['chisq_array'] = [float(_) for _ in outp['chisq_array'][0]]
        ret['npix_array'] = [int(_) for _ in outp['npix_array'][0]]
    ret['nm_vel'] = nm_vel if is_batch else float(nm_vel[0].item())
    ret['nm_nit'] = nit if is_batch else int(nit[0].item())
    ret['nm_nfev'] = nfev if is_batch else int(nfev[0].item())
    ret['nm_rounds'] = nmstats['rounds']
    ret['objective_evals'] = nmstats['evals'] + obj.nfev
    # rows the lock-step Nelder-Mead launched (an upper bound known on the host);
    # the rows behind the device counts are skipped by the objective kernel
    ret['nm_launched_rows'] = slots
    ret['second_minimizer_run'] = second_run
    if bfgs_info is not None:
        ret['bfgs'] = bfgs_info
    ret['optimizer_run'] = True
    return ret


def _lib_nonfinite():
    from . import _lib
    return _lib.ST_NONFINITE
