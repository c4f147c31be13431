"""Template interpolators -- API mirror of py/rvspecfit/spec_inter.py.

`getInterpolator(setup, config)` returns a `SpecInterpolator` with the same
attributes the reference exposes (.eval, .outsideFlag, .lam, .parnames,
.log_step, .revision); underneath sits a device-resident
`library.TemplateLibrary` and the HIP polylinear / NN kernels.
"""
import os

import numpy as np
import torch

from .library import TemplateLibrary

LIB_NPZ_NAME = 'rvsgpu_%s.npz'  # converted artefact of one setup


class interp_cache:
    """process-wide cache, as spec_inter.interp_cache (spec_inter.py:289-293)"""
    interps = {}
    template_lib = None
    registered = {}


def register_library(lib, template_lib=None):
    """Make an in-memory TemplateLibrary visible to getInterpolator."""
    interp_cache.registered[(template_lib, lib.name)] = lib
    interp_cache.interps.pop(lib.name, None)


class SpecInterpolator:

    def __init__(self, lib, filename=''):
        self.lib = lib
        self.name = lib.name
        self.lam = lib.lam
        self.parnames = lib.parnames
        self.log_step = lib.log_step
        self.revision = lib.revision
        self.filename = filename
        self.creation_soft_version = getattr(lib, 'creation_soft_version', '')
        self.objid = hash((self.name, self.parnames, self.revision, filename))

    def __hash__(self):
        return self.objid

    def _vec(self, param0):
        if isinstance(param0, dict):
            try:
                param0 = [param0[_] for _ in self.parnames]
            except KeyError as exc:
                raise ValueError(f'The parameter {exc.args[0]} not found. '
                                 'Required list of parameters is: ' +
                                 ','.join(self.parnames))
        return torch.as_tensor(np.asarray(param0, dtype=np.float64)[None, :]
                               ).to(self.lib.device)

    def eval(self, param0):
        """Evaluate the spectrum at a given parameter (numpy float64 [ntp])."""
        templ, _ = self.lib.eval_batch(self._vec(param0))
        return templ[0].cpu().numpy()

    def outsideFlag(self, param0):
        _, outside = self.lib.eval_batch(self._vec(param0))
        return float(outside[0].item())

    def eval_batch(self, params):
        """params [J, ndim] (numpy or device tensor) -> device tensors
        (templ [J, ntp], outside [J])"""
        if not isinstance(params, torch.Tensor):
            params = torch.as_tensor(np.asarray(params, dtype=np.float64))
        return self.lib.eval_batch(params.to(self.lib.device))


def getInterpolator(HR, config, warmup_cache=False, cache=None):
    """spec_inter.getInterpolator (spec_inter.py:296-398)."""
    if cache is None:
        if config['template_lib'] != interp_cache.template_lib:
            interp_cache.template_lib = config['template_lib']
            interp_cache.interps = {}
        cache = interp_cache.interps
    if HR not in cache:
        tl = config['template_lib']
        lib = interp_cache.registered.get((tl, HR)) or \
            interp_cache.registered.get((None, HR))
        fname = ''
        if lib is None:
            fname = os.path.join(tl, LIB_NPZ_NAME % HR)
            if not os.path.exists(fname):
                raise RuntimeError(
                    'No converted template library %s (run '
                    'tools/convert_artefacts.py on the rvspecfit template '
                    'directory)' % fname)
            lib = TemplateLibrary.from_npz(HR, fname)
        cache[HR] = SpecInterpolator(lib, filename=fname)
    return cache[HR]


def getSpecParams(setup, config):
    return getInterpolator(setup, config).parnames


def get_libs(names, config):
    return {n: getInterpolator(n, config).lib for n in names}
