"""Template evaluator plug-in for an UNMODIFIED rvspecfit.

The reference has one sanctioned hook for foreign template evaluators:
`interpolation_type = 'generic'` in `interp_<setup>.h5` makes
spec_inter.getInterpolator import `fd['module']` and build

    interper = getattr(mod, fd['class_name'])(fd)
    extraper = getattr(mod, fd['outside_class_name'])(fd)

(spec_inter.py:371-378; the record the NN trainer writes is
nn/train_interpolator.py:373-393; getInterpolator adds fd['template_lib']).
Both objects are called with the MAPPED parameter vector --
SpecInterpolator.eval / .outsideFlag apply `mapper.forward` first
(spec_inter.py:257-286) -- `interper(p)` returns the float64 template
[len(lam)], `extraper(p)` the outside flag (0 inside the grid).

`Evaluator` / `Outside` below are those two classes on top of the MI355X
template kernels (rvs_template_polylinear / _tri / _nn): with
`record(...)` saved as `interp_<setup>.h5` (serializer.save_dict_to_hdf5) next
to the converted artefact `rvsgpu_<setup>.npz`, every `getCurTempl` of the
reference evaluates its templates on the GPU; everything else stays the
reference's code.  There is no CPU fallback: without the HIP library or a GPU the
constructor raises.
"""
import os

import numpy as np
import torch

from . import _lib
from .library import TemplateLibrary
from .spec_inter import LIB_NPZ_NAME

_LIBS = {}


def _library(fd):
    """one device-resident TemplateLibrary per converted artefact and process"""
    _lib.require_gpu()
    fname = fd.get('rvsgpu_file')
    if fname is None:
        fname = LIB_NPZ_NAME % fd['setup']
    if not os.path.isabs(fname):
        fname = os.path.join(fd['template_lib'], fname)
    if fname not in _LIBS:
        if not os.path.exists(fname):
            raise RuntimeError(
                'No converted template library %s (run '
                'tools/convert_artefacts.py on the rvspecfit template '
                'directory)' % fname)
        _LIBS[fname] = TemplateLibrary.from_npz(str(fd.get('setup', 'generic')),
                                                fname)
    return _LIBS[fname]


def _mapped(lib, p):
    p = np.asarray(p, dtype=np.float64).reshape(1, -1)
    if p.shape[1] != lib.ndim:
        raise ValueError('expected %d mapped parameters, got %d'
                         % (lib.ndim, p.shape[1]))
    return torch.as_tensor(p).to(lib.device)


class Evaluator:
    """`interper` of a 'generic' setup: mapped parameter vector -> template"""

    def __init__(self, fd):
        self.lib = _library(fd)

    def __call__(self, p):
        templ, _ = self.lib.eval_batch(_mapped(self.lib, p), mapped=True)
        return templ[0].cpu().numpy()

    def batch(self, P):
        """[J, ndim] mapped vectors -> device tensors (templ [J, ntp], outside)"""
        P = torch.as_tensor(np.asarray(P, dtype=np.float64)).to(self.lib.device)
        return self.lib.eval_batch(P, mapped=True)


class Outside:
    """`extraper` of a 'generic' setup: mapped parameter vector -> outside flag"""

    def __init__(self, fd):
        self.lib = _library(fd)

    def __call__(self, p):
        _, outside = self.lib.eval_batch(_mapped(self.lib, p), mapped=True)
        return float(outside[0].item())


def record(setup, npz_path, mapper_module='rvspecfit.read_grid',
           mapper_class_name='LogParamMapper'):
    """The dict to save as interp_<setup>.h5 (serializer.save_dict_to_hdf5) so
    that the reference loads this plug-in for `setup`.  The parameter mapper
    stays the reference's own class (LogParamMapper for grids, the nn Mapper
    for MLP setups: pass mapper_module='rvspecfit.nn.NNInterpolator',
    mapper_class_name='Mapper')."""
    d = np.load(npz_path, allow_pickle=False)
    log_ids = [int(_) for _ in np.atleast_1d(d['log_ids'])]
    if 'nn_dims' in d.files:
        mapper_args = (np.asarray(d['nn_M']), np.asarray(d['nn_S']), log_ids)
    else:
        mapper_args = (log_ids, )
    return {
        'interpolation_type': 'generic',
        'module': 'rvspecfit_amd.plugin',
        'class_name': 'Evaluator',
        'outside_class_name': 'Outside',
        'setup': setup,
        'rvsgpu_file': os.path.basename(npz_path),
        'mapper_module': mapper_module,
        'mapper_class_name': mapper_class_name,
        'mapper_args': mapper_args,
        'parnames': tuple(str(_) for _ in d['parnames']),
        'lam': np.asarray(d['lam'], dtype=np.float64),
        'log_spec': True,
        'log_step': bool(d['log_step']),
        'revision': str(d['revision']) if 'revision' in d.files else '',
    }
