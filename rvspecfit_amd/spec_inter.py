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


def _evaluator_library(name, ntp, ndim, extra):
    """a TemplateLibrary for an evaluator object that is handed its arrays (the
    classes below): the wavelength grid plays no part in the evaluation"""
    d = dict(lam=np.arange(1, ntp + 1, dtype=np.float64), log_step=False,
             log_ids=np.zeros(0, dtype=np.int64),
             parnames=['p%d' % i for i in range(ndim)])
    d.update(extra)
    return TemplateLibrary(name, d)


def _one_mapped(lib, p):
    p = np.asarray(p, dtype=np.float64).reshape(1, lib.ndim)
    return lib.eval_batch(torch.as_tensor(p).to(lib.device), mapped=True)


class GridInterp:
    """spec_inter.GridInterp (spec_inter.py:95-194): polylinear interpolation of the
    2^ndim spectra at the corners of the grid cell of p -- p in the grid's own
    (mapped) coordinates -- and the nearest spectrum in ptp-scaled coordinates when
    p is off the grid or a corner of its cell has no spectrum (idgrid -1); a p
    that is not finite gives the first spectrum.  rvs_template_polylinear
    (csrc/template.hip) on arrays resident in HBM.
    uvecs: the ndim axes; idgrid: int [len(axis 0), ...] row of `dats` at every
    node or -1; vecs [ndim, n]: the nodes' coordinates; dats [n, npix] float32;
    exp: the rows are logarithms."""

    def __init__(self, uvecs, idgrid, vecs, dats, exp=True):
        dats = np.asarray(dats)
        if dats.dtype != np.float32:
            raise TypeError('GridInterp: the kernel blends float32 rows (what '
                            'interpdat_<setup>.npy holds), got %s' % dats.dtype)
        self.uvecs = [np.asarray(u, dtype=np.float64) for u in uvecs]
        self.idgrid = np.asarray(idgrid)
        self.exp = bool(exp)
        self.ndim = len(self.uvecs)
        self.lens = np.array([len(u) for u in self.uvecs])
        self.ptp = np.ptp(np.asarray(vecs, dtype=np.float64), axis=1)
        extra = dict(dats=dats, idgrid=self.idgrid, vec=vecs, log_spec=self.exp)
        for i, u in enumerate(self.uvecs):
            extra['uvec%d' % i] = u
        self.lib = _evaluator_library('GridInterp', dats.shape[1], self.ndim,
                                      extra)
        self._vecs = np.asarray(vecs, dtype=np.float64)
        self._nearest_lib = None

    def get_nearest(self, p):
        """row of the spectrum nearest to p (ptp-scaled Euclidean distance)"""
        if self._nearest_lib is None:
            # (the same nodes under a grid without cells: every p takes the
            # nearest-neighbour branch; no spectra needed)
            extra = dict(dats=np.zeros((self._vecs.shape[1], 4), dtype=np.float32),
                         idgrid=np.full(self.idgrid.shape, -1, dtype=np.int64),
                         vec=self._vecs, log_spec=False)
            for i, u in enumerate(self.uvecs):
                extra['uvec%d' % i] = u
            self._nearest_lib = _evaluator_library('GridInterp.nearest', 4,
                                                   self.ndim, extra)
        _, _, cell, _ = self._nearest_lib.eval_batch(
            torch.as_tensor(np.asarray(p, dtype=np.float64).reshape(
                1, self.ndim)).to(self.lib.device), details=True, mapped=True)
        return int(cell[0, 1].item())

    def __call__(self, p):
        return _one_mapped(self.lib, p)[0][0].cpu().numpy()

    def batch(self, P):
        """P [J, ndim] (mapped) -> device tensor [J, npix]"""
        P = torch.as_tensor(np.asarray(P, dtype=np.float64))
        return self.lib.eval_batch(P.to(self.lib.device), mapped=True)[0]


class GridOutsideCheck:
    """spec_inter.GridOutsideCheck (spec_inter.py:62-92): 0 for a p inside a grid
    cell whose corners all carry spectra, else the ptp-scaled distance to the
    nearest spectrum (the penalty scale of get_chisq)."""

    def __init__(self, uvecs, vecs, idgrid):
        self.uvecs = [np.asarray(u, dtype=np.float64) for u in uvecs]
        self.idgrid = np.asarray(idgrid)
        self.ndim = len(self.uvecs)
        self.lens = np.array([len(u) for u in self.uvecs])
        self.Ns = self.idgrid.shape
        self.ptp = np.ptp(np.asarray(vecs, dtype=np.float64), axis=1)
        n = np.asarray(vecs).shape[1]
        # (the check reads no spectrum: one column stands for the rows)
        extra = dict(dats=np.zeros((n, 4), dtype=np.float32), idgrid=self.idgrid,
                     vec=vecs, log_spec=False)
        for i, u in enumerate(self.uvecs):
            extra['uvec%d' % i] = u
        self.lib = _evaluator_library('GridOutsideCheck', 4, self.ndim, extra)

    def __call__(self, p):
        out = float(_one_mapped(self.lib, p)[1][0].item())
        return out if out != 0 else 0


class TriInterp:
    """spec_inter.TriInterp (spec_inter.py:11-59): barycentric blend of the ndim + 1
    spectra of the Delaunay simplex that holds p; NaN outside the hull.
    rvs_template_tri (csrc/template.hip); `triang` is a scipy.spatial.Delaunay."""

    def __init__(self, triang, dats, exp=True):
        self.triang = triang
        self.dats = np.asarray(dats, dtype=np.float64)
        self.exp = bool(exp)
        self.ndim = triang.ndim
        extra = dict(simplices=triang.simplices, transform=triang.transform,
                     extraflags=np.zeros(len(triang.simplices)), dats=self.dats,
                     log_spec=self.exp)
        self.lib = _evaluator_library('TriInterp', self.dats.shape[1], self.ndim,
                                      extra)

    def __call__(self, p):
        templ, _, sx, _ = self.lib.eval_batch(
            torch.as_tensor(np.asarray(p, dtype=np.float64).reshape(
                1, self.ndim)).to(self.lib.device), details=True, mapped=True)
        if int(sx[0].item()) >= len(self.triang.simplices):   # no simplex holds p
            return np.nan
        spec = templ[0].cpu().numpy()
        return float(spec[0]) if spec.size == 1 else spec


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
