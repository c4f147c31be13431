"""Cross-correlation first guess -- API mirror of py/rvspecfit/fitter_ccf.py."""
import logging

import numpy as np
import torch

from . import _lib
from . import engine
from . import spec_inter
from .spec_fit import SpecData, as_batch


class CCFCache:
    """fitter_ccf.CCFCache (fitter_ccf.py:13-18): what get_ccf_info has handed out,
    by setup.  (The arrays the kernels read live with the TemplateLibrary in HBM;
    these are the host views of the same set.)"""
    ccf_info = {}
    ccfs = {}
    ccf2s = {}
    ccf_models = {}


def get_ccf_info(spec_setup, config):
    """fitter_ccf.get_ccf_info (fitter_ccf.py:21-59): (fft, fft2, models, info) of
    the CCF template set of a setup -- complex128 [T, N/2 + 1] transforms of the
    prepared templates and of their squares, the prepared templates themselves
    [T, N] (None if the library was converted without them), and the dictionary
    make_ccf saved: params [T, ndim], vsinis [T], parnames, ccfconf."""
    if spec_setup not in CCFCache.ccfs:
        lib = spec_inter.get_libs([spec_setup], config)[spec_setup]
        cc = lib.ccf_set(config)
        T = cc['params'].shape[0]
        nr = cc['npoints'] // 2 + 1

        def cplx(t):
            return t.cpu().numpy().reshape(-1).view(np.complex128).reshape(T, nr)
        conf = dict(logl0=cc['logl0'], logl1=cc['logl1'], npoints=cc['npoints'],
                    continuum=cc['continuum'], maxcontpts=cc['maxcontpts'])
        if cc['splinestep'] is not None:
            conf['splinestep'] = cc['splinestep']
        CCFCache.ccfs[spec_setup] = cplx(cc['fft'])
        CCFCache.ccf2s[spec_setup] = cplx(cc['fft2'])
        CCFCache.ccf_models[spec_setup] = cc['mod']
        CCFCache.ccf_info[spec_setup] = dict(
            params=cc['params'], vsinis=cc['vsinis'],
            parnames=tuple(lib.parnames), ccfconf=conf)
    return (CCFCache.ccfs[spec_setup], CCFCache.ccf2s[spec_setup],
            CCFCache.ccf_models[spec_setup], CCFCache.ccf_info[spec_setup])


def fit(specdata, config):
    """fitter_ccf.fit (fitter_ccf.py:62-253).

    One spectrum (SpecData or list of SpecData): the reference's result dict
    (best_par dict, best_vel, best_ccf, best_vsini, best_model {arm: array},
    proc_spec {arm: array}, vel_grid) as numpy / floats; raises RuntimeError
    when the cross-correlation fails.
    SpecBatch: device tensors with a leading S axis (best_id, best_par [S,ndim],
    best_vel [S], best_ccf [S,nvel], best_vsini [S], status [S], proc_spec)."""
    batch, is_batch = as_batch(specdata)
    libs = spec_inter.get_libs(batch.names, config)
    r = engine.ccf_fit(batch, libs, config)
    ref = libs[batch.names[0]].ccf_set(config)
    best_id = r['best_id']
    par = ref['params_dev'][best_id]
    vs = ref['vsinis_dev'][best_id]
    if is_batch:
        return dict(best_id=best_id, best_par=par, best_vel=r['best_vel'],
                    best_ccf=r['best_ccf'], best_vsini=vs, status=r['status'],
                    proc_spec=dict(zip(batch.names, r['proc_spec'])),
                    proc_ivar=dict(zip(batch.names, r['proc_ivar'])),
                    vel_grid=r['vel_grid'])
    st = int(r['status'][0].item())
    if st & _lib.ST_CCF_FAILED:
        logging.error('Cross-correlation failed')
        raise RuntimeError('Cross-correlation step failed')
    bid = int(best_id[0].item())
    best_vel = float(r['best_vel'][0].item())
    parnames = libs[batch.names[0]].parnames
    best_model = {}
    for name, step in zip(batch.names, r['steps']):
        mod = libs[name].ccf_set(config)['mod']
        if mod is not None:
            best_model[name] = np.roll(mod[bid], int(best_vel / step))
    v = ref['vsinis'][bid]
    return dict(best_par=dict(zip(parnames, ref['params'][bid])),
                best_vel=best_vel,
                best_ccf=r['best_ccf'][0].cpu().numpy(),
                best_vsini=None if np.isnan(v) else float(v),
                best_model=best_model,
                proc_spec={n: p[0].cpu().numpy()
                           for n, p in zip(batch.names, r['proc_spec'])},
                vel_grid=r['vel_grid_host'])
