"""Cross-correlation first guess -- API mirror of py/rvspecfit/fitter_ccf.py."""
import logging

import numpy as np
import torch

from . import _lib
from . import engine
from . import spec_inter
from .spec_fit import SpecData, as_batch


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
