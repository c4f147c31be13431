"""Device-resident template library of one spectral setup (arm).

Holds, as PyTorch-ROCm tensors in HBM, exactly what the reference keeps in its
process-wide caches (spec_inter.interp_cache, fitter_ccf.CCFCache):

  interp_%s.h5 / interpdat_%s.npy   -> dats (float32 [ngrid, ntp], log flux),
        idgrid, uvecs, vec (mapped grid points), lam, log_step, parnames, mapper
  ccf_%s.h5 / ccfdat_%s.npz / ccfmod_%s.npy -> fft, fft2 (complex128
        [T, nfft/2+1]), params, vsinis, ccfconf, model (kept on the host: it is
        only rolled and returned, fitter_ccf.py:238-241)
  NN checkpoint + 'generic' record  -> float32 weights, Mapper M/S, hull

The on-disk form read here is a plain .npz ("converted artefact", same keys as
tests/golden/lib_*.npz); h5py is not available next to torch in this image, so
reference artefact directories are converted once with tools/convert_artefacts.py
under an interpreter that has h5py.
"""
import numpy as np
import torch

from . import _lib


def _dev(a, dtype, device):
    if isinstance(a, torch.Tensor):          # already resident (synthetic libraries)
        return a.to(device=device, dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(device)


# find_simplex of Delaunay libraries: bucket grid from this many simplices up
TRI_BUCKET_MIN = 256
TRI_BUCKETS = True
TRI_WIDE_CELLS = 4096   # a simplex whose box overlaps more cells goes to the wide list
_TRI_SHARED = {}        # device arrays of the (few) distinct triangulations in use


def tri_buckets(transform, ndim, per_cell=8, max_n=48, max_entries=1 << 25,
                wide_cells=4096):
    """The bucket grid of rvs_template_tri_buckets for a Delaunay triangulation given
    as scipy's `transform` [nsimplex, ndim + 1, ndim] (spec_inter.py:11-59 keeps the
    Delaunay object; the artefact reader exports its arrays): vertices from the
    transform (the last one is r, the others r + the columns of the inverse), bounding
    boxes grown by 1e-9 of the extent, a uniform grid of n^ndim cells with about
    `per_cell` simplices per cell, every simplex entered in each cell its box
    overlaps, the lists ascending.  A simplex whose box overlaps more than
    `wide_cells` cells (the long slivers Qhull leaves on a hull) is entered ONCE, in a
    list of its own behind the last cell -- list number ncell --, which every query
    tests as well.  Degenerate simplices (non-finite transform: they never pass the
    inside test) are in no list.  Returns dict(cell_start int32 [ncell + 2],
    cell_list int32 [entries], lo, inv_w float64 [ndim], n int32 [ndim])."""
    tr = np.asarray(transform, dtype=np.float64)
    ns = tr.shape[0]
    good = np.isfinite(tr).all(axis=(1, 2))
    if good.any():
        with np.errstate(all='ignore'):
            det = np.linalg.det(tr[good][:, :ndim, :])
        g2 = np.zeros(ns, dtype=bool)
        g2[np.nonzero(good)[0][np.isfinite(det) & (det != 0)]] = True
        good = g2
    ids = np.nonzero(good)[0]
    if len(ids) == 0:
        return dict(cell_start=np.zeros(3, dtype=np.int32),
                    cell_list=np.zeros(1, dtype=np.int32), lo=np.zeros(ndim),
                    inv_w=np.zeros(ndim), n=np.ones(ndim, dtype=np.int32))
    r = tr[ids, ndim, :]
    cols = np.linalg.inv(tr[ids, :ndim, :])          # column j = vertex j - r
    verts = np.concatenate([r[:, None, :] + np.swapaxes(cols, 1, 2),
                            r[:, None, :]], axis=1)   # [n, ndim + 1, ndim]
    blo, bhi = verts.min(axis=1), verts.max(axis=1)
    lo, hi = blo.min(axis=0), bhi.max(axis=0)
    ext = np.where(hi > lo, hi - lo, 1.0)
    m = 1e-9 * ext
    n1 = int(np.clip(round((len(ids) / float(per_cell))**(1.0 / ndim)), 1, max_n))
    while True:
        n = np.full(ndim, n1, dtype=np.int64)
        inv_w = n / ext
        i0 = np.clip(np.floor((blo - m - lo) * inv_w), 0, n - 1).astype(np.int64)
        i1 = np.clip(np.floor((bhi + m - lo) * inv_w), 0, n - 1).astype(np.int64)
        span = i1 - i0 + 1
        cnt = span.prod(axis=1)
        wide = cnt > wide_cells
        total = int(cnt[~wide].sum())
        if total <= max_entries or n1 == 1:
            break
        n1 = max(1, n1 // 2)
    # every (simplex, cell) pair of the narrow simplices, without a Python loop: entry e
    # belongs to simplex owner[e] and is cell number local[e] of its box, decoded digit
    # by digit in the box's own mixed radix
    nar = np.nonzero(~wide)[0]
    c_n = cnt[nar]
    owner = np.repeat(nar, c_n)
    local = np.arange(int(c_n.sum()), dtype=np.int64) - np.repeat(
        np.cumsum(c_n) - c_n, c_n)
    digits = []
    for d in range(ndim - 1, -1, -1):
        sp = span[owner, d]
        digits.append(i0[owner, d] + local % sp)
        local = local // sp
    cells = np.zeros(len(owner), dtype=np.int64)
    for d, dg in zip(range(ndim), digits[::-1]):
        cells = cells * n[d] + dg
    sims = ids[owner]
    order = np.lexsort((sims, cells))        # by cell, ascending simplex inside
    cells, sims = cells[order], sims[order]
    ncell = int(n.prod())
    start = np.zeros(ncell + 2, dtype=np.int64)
    np.add.at(start, cells + 1, 1)
    start[ncell + 1] = int(wide.sum())
    start = np.cumsum(start)
    lst = np.concatenate([sims, np.sort(ids[wide])])
    if len(lst) == 0:
        lst = np.zeros(1, dtype=np.int64)
    return dict(cell_start=start.astype(np.int32), cell_list=lst.astype(np.int32),
                lo=lo, inv_w=inv_w, n=n.astype(np.int32))


class TemplateLibrary:

    def __init__(self, name, d, device='cuda'):
        self.name = name
        self.device = device
        d = dict(d)
        self.lam = np.ascontiguousarray(d['lam'], dtype=np.float64)
        self.ntp = len(self.lam)
        self.log_step = bool(d['log_step'])
        self.log_ids = [int(_) for _ in np.atleast_1d(d.get('log_ids', [0]))]
        self.parnames = tuple(str(_) for _ in d['parnames'])
        self.ndim = len(self.parnames)
        self.log_mask = 0
        for i in self.log_ids:
            self.log_mask |= (1 << i)
        self.revision = str(d.get('revision', ''))
        self.creation_soft_version = str(d.get('creation_soft_version', ''))
        self.knots = _dev(self.lam, torch.float64, device)
        self.knots3 = np.ascontiguousarray(self.lam[:3])
        self.lnstep = float(np.log(self.lam[1] / self.lam[0]))
        # rvs_spline_construct form: 1 = power-form records; | 2 = windowed
        # solve, allowed when neighbouring knot spacings agree to ~1 %
        hh = np.diff(self.lam)
        near_uniform = bool(np.all(np.abs(hh[1:] / hh[:-1] - 1) < 5e-3))
        self.spline_form = 3 if near_uniform else 1
        self.spline_factors = None
        if near_uniform and str(device) != 'cpu':
            self.spline_factors = torch.empty(
                _lib.lib().rvs_spline_factors_len(self.ntp), dtype=torch.float64,
                                              device=device)
            rc = _lib.lib().rvs_spline_factors(
                _lib.ptr(self.knots), self.ntp, _lib.ptr(self.spline_factors),
                _lib.stream())
            _lib.check(rc, 'rvs_spline_factors')
        self.kind = 'regulargrid'
        if 'simplices' in d:
            # interpolation_type 'triangulation' (make_nd without --regulargrid):
            # the reference's Delaunay object, exported as arrays
            self.kind = 'triangulation'
            # one device copy per distinct triangulation: the arms of a setup are
            # computed on one parameter grid, and find_simplex is then done once for
            # all of them (rvs_nm_run: arms with the same arrays share the ids)
            import hashlib
            tkey = (str(device), hashlib.sha1(
                np.ascontiguousarray(d['transform']).tobytes()).hexdigest(),
                hashlib.sha1(np.ascontiguousarray(d['simplices']).tobytes()).hexdigest())
            shared = _TRI_SHARED.get(tkey)
            if shared is None:
                shared = _TRI_SHARED[tkey] = dict(
                    simplices=_dev(d['simplices'], torch.int32, device),
                    transform=_dev(d['transform'], torch.float64, device))
                while len(_TRI_SHARED) > 8:
                    _TRI_SHARED.pop(next(iter(_TRI_SHARED)))
            self._tri_shared = shared
            self.tri_simplices = shared['simplices']
            self.tri_transform = shared['transform']
            self.tri_extraflags = _dev(np.asarray(d['extraflags']).reshape(-1),
                                       torch.float64, device)
            self.tri_nsimplex = int(np.asarray(d['simplices']).shape[0])
            # find_simplex through a bucket grid (rvs_template_tri_buckets) from a
            # few hundred simplices up; TRI_BUCKETS = False keeps the exhaustive
            # search (tests hold one against the other)
            self._tri_bk = self._tri_keep = None
            if self.tri_nsimplex >= TRI_BUCKET_MIN and str(device) != 'cpu':
                bkey = ('buckets', TRI_WIDE_CELLS)
                if bkey not in shared:
                    bk = tri_buckets(np.asarray(d['transform']), self.ndim,
                                     wide_cells=TRI_WIDE_CELLS)
                    shared[bkey] = (bk, _dev(bk['cell_start'], torch.int32, device),
                                    _dev(bk['cell_list'], torch.int32, device))
                bk = shared[bkey][0]
                self.tri_nwide = int(bk['cell_start'][-1] - bk['cell_start'][-2])
                self._tri_keep = shared[bkey][1:]
                t = _lib.TriBuckets()
                t.cell_start = self._tri_keep[0].data_ptr()
                t.cell_list = self._tri_keep[1].data_ptr()
                for k in range(self.ndim):
                    t.lo[k], t.inv_w[k] = float(bk['lo'][k]), float(bk['inv_w'][k])
                    t.n[k] = int(bk['n'][k])
                self._tri_bk = t
            self.dats = _dev(d['dats'], torch.float64, device)
            self.exp_flag = int(bool(d.get('log_spec', True)))
        elif 'dats' in d:
            idgrid = np.asarray(d['idgrid'], dtype=np.int64)
            self.lens = np.array(idgrid.shape, dtype=np.int32)
            uvecs = [np.asarray(d['uvec%d' % i], dtype=np.float64)
                     for i in range(self.ndim)]
            self.uvecs_host = uvecs
            vec = np.asarray(d['vec'], dtype=np.float64)  # [ndim, ngrid] mapped
            ptp = np.ptp(vec, axis=1)
            self.ptp = np.ascontiguousarray(ptp)
            self.ngrid = vec.shape[1]
            self.dats = _dev(d['dats'], torch.float32, device)
            self.idgrid = _dev(idgrid.ravel(), torch.int64, device)
            self.uvecs = _dev(np.concatenate(uvecs), torch.float64, device)
            self.vecs_s = _dev((vec.T / ptp[None, :]), torch.float64, device)
            self.exp_flag = int(bool(d.get('log_spec', True)))
        if 'nn_dims' in d:
            self.kind = 'nn'
            self.nn_dims = np.asarray(d['nn_dims'], dtype=np.int32)
            nl = len(self.nn_dims) - 1
            self.nn_W = [_dev(d['nn_W%d' % i], torch.float32, device)
                         for i in range(nl)]
            self.nn_b = [_dev(d['nn_b%d' % i], torch.float32, device)
                         for i in range(nl)]
            self.nn_M = _dev(d['nn_M'], torch.float64, device)
            self.nn_S = _dev(d['nn_S'], torch.float64, device)
            self.nn_hull = None
            if 'nn_pts' in d:
                # OutsideInterpolator.__init__ (nn/RVSInterpolator.py:47-61):
                # facet equations of the two 2-D convex hulls (host, once)
                import scipy.spatial
                pts = np.asarray(d['nn_pts'], dtype=np.float64)
                self.nn_hull = (scipy.spatial.ConvexHull(pts[:, :2]).equations,
                                scipy.spatial.ConvexHull(pts[:, 2:]).equations)
        # the two CCF template sets of a setup: continuum-normalised (files
        # ccf_<setup>.h5 ..., keys ccf_*) and not (rvs_make_ccf --nocontinuum,
        # files ccf_nocont_<setup>.h5 ..., keys ccfnc_*); make_ccf.py:19-36
        self.ccf_sets = {}
        for pre in ('ccf_', 'ccfnc_'):
            if pre + 'fft' not in d:
                continue
            # the set is what its own ccfconf['continuum'] says it is
            # (make_ccf.py:483-493), whatever prefix an older converter gave it
            cont = bool(d[pre + 'continuum'])
            if cont in self.ccf_sets:
                raise ValueError(
                    'setup %s: two CCF template sets with continuum=%s; '
                    'reconvert the artefacts with tools/convert_artefacts.py'
                    % (name, cont))
            fft = np.ascontiguousarray(d[pre + 'fft'], dtype=np.complex128)
            fft2 = np.ascontiguousarray(d[pre + 'fft2'], dtype=np.complex128)
            self.ccf_sets[cont] = dict(
                T=fft.shape[0],
                nfft=int(d[pre + 'npoints']),
                fft=_dev(fft.view(np.float64), torch.float64, device),
                fft2=_dev(fft2.view(np.float64), torch.float64, device),
                mod=np.asarray(d[pre + 'mod']) if pre + 'mod' in d else None,
                params=np.asarray(d[pre + 'params'], dtype=np.float64),
                vsinis=np.asarray(d[pre + 'vsinis'], dtype=np.float64),
                params_dev=_dev(d[pre + 'params'], torch.float64, device),
                vsinis_dev=_dev(np.nan_to_num(np.asarray(d[pre + 'vsinis'],
                                                         dtype=np.float64),
                                              nan=0.0), torch.float64, device),
                logl0=float(d[pre + 'logl0']), logl1=float(d[pre + 'logl1']),
                npoints=int(d[pre + 'npoints']),
                continuum=bool(d[pre + 'continuum']),
                splinestep=float(d[pre + 'splinestep'])
                if pre + 'splinestep' in d else None,
                maxcontpts=int(d[pre + 'maxcontpts'])
                if pre + 'maxcontpts' in d else 20)
        self.ccf = self.ccf_sets.get(True)

    def ccf_set(self, config):
        """The CCF template set config['ccf_continuum_normalize'] selects
        (get_ccf_info, fitter_ccf.py:40-47: missing / None = the
        continuum-normalised one)."""
        cont = (config or {}).get('ccf_continuum_normalize')
        cont = True if cont is None else bool(cont)
        if cont not in self.ccf_sets:
            raise RuntimeError(
                'setup %s has no %s CCF template set (ccf_%s%s.h5 was not '
                'converted)' % (self.name, 'continuum-normalised' if cont
                                else 'non-normalised',
                                '' if cont else 'nocont_', self.name))
        return self.ccf_sets[cont]

    @classmethod
    def from_npz(cls, name, path, device='cuda'):
        return cls(name, np.load(path, allow_pickle=False), device=device)

    # -- A3 / A4 : template evaluation for a batch of parameter vectors ------
    def eval_batch(self, params, details=False, mapped=False):
        """params float64 [J, ndim] (device) -> templ [J, ntp], outside [J].
        mapped=True: `params` are already what the setup's parameter mapper
        returns (LogParamMapper.forward, read_grid.py:127-145; nn Mapper.forward,
        nn/NNInterpolator.py:159-171) -- the contract of the reference's
        evaluator classes (spec_inter.py:257-286), used by plugin.py."""
        L = _lib.lib()
        J = params.shape[0]
        params = params.to(torch.float64).contiguous()
        log_mask = 0 if mapped else self.log_mask
        templ = torch.empty((J, self.ntp), dtype=torch.float64,
                            device=self.device)
        outside = torch.empty(J, dtype=torch.float64, device=self.device)
        if self.kind == 'nn':
            return self._eval_nn(params, templ, outside, mapped)
        if self.kind == 'triangulation':
            sx = torch.empty(J, dtype=torch.int32, device=self.device)
            wts = torch.zeros((J, self.ndim + 1), dtype=torch.float64,
                              device=self.device) if details else None
            rc = self._tri_call(log_mask, params, J, templ, outside, sx, wts,
                                _lib.stream())
            _lib.check(rc, 'rvs_template_tri')
            if details:
                return templ, outside, sx, wts
            return templ, outside
        nv = 1 << self.ndim
        cell = wts = None
        if details:
            cell = torch.zeros((J, 2 + nv), dtype=torch.int32,
                               device=self.device)
            wts = torch.zeros((J, nv), dtype=torch.float64, device=self.device)
        rc = L.rvs_template_polylinear(
            _lib.ptr(self.dats), self.ngrid, self.ntp, _lib.ptr(self.idgrid),
            _lib.ptr(self.uvecs), _lib.ptr(self.lens), self.ndim,
            _lib.ptr(self.vecs_s), _lib.ptr(self.ptp), log_mask,
            self.exp_flag, _lib.ptr(params), J, _lib.ptr(templ),
            _lib.ptr(outside), _lib.ptr(cell), _lib.ptr(wts), _lib.stream())
        _lib.check(rc, 'rvs_template_polylinear')
        if details:
            return templ, outside, cell, wts
        return templ, outside

    def _tri_call(self, log_mask, params, J, templ, outside, sx, wts, stream):
        import ctypes
        L = _lib.lib()
        if self._tri_bk is not None and TRI_BUCKETS:
            return L.rvs_template_tri_buckets(
                _lib.ptr(self.dats), self.ntp, _lib.ptr(self.tri_simplices),
                _lib.ptr(self.tri_transform), _lib.ptr(self.tri_extraflags),
                self.tri_nsimplex, self.ndim, log_mask, self.exp_flag,
                ctypes.addressof(self._tri_bk), _lib.ptr(params), J,
                _lib.ptr(templ), _lib.ptr(outside), _lib.ptr(sx), _lib.ptr(wts),
                stream)
        return L.rvs_template_tri(
            _lib.ptr(self.dats), self.ntp, _lib.ptr(self.tri_simplices),
            _lib.ptr(self.tri_transform), _lib.ptr(self.tri_extraflags),
            self.tri_nsimplex, self.ndim, log_mask, self.exp_flag,
            _lib.ptr(params), J, _lib.ptr(templ), _lib.ptr(outside),
            _lib.ptr(sx), _lib.ptr(wts), stream)

    def eval_into(self, params, J, templ, outside, stream, scratch=None):
        """launch-only variant of eval_batch on caller-owned buffers (no
        allocation, no synchronisation): params [>=J, ndim], templ [>=J, ntp],
        outside [>=J]; scratch: int32 [>=J] for triangulation libraries, a dict
        of activation buffers + the torch stream for nn libraries."""
        L = _lib.lib()
        if self.kind == 'regulargrid':
            rc = L.rvs_template_polylinear(
                _lib.ptr(self.dats), self.ngrid, self.ntp, _lib.ptr(self.idgrid),
                _lib.ptr(self.uvecs), _lib.ptr(self.lens), self.ndim,
                _lib.ptr(self.vecs_s), _lib.ptr(self.ptp), self.log_mask,
                self.exp_flag, _lib.ptr(params), J, _lib.ptr(templ),
                _lib.ptr(outside), None, None, stream)
            _lib.check(rc, 'rvs_template_polylinear')
        elif self.kind == 'triangulation':
            rc = self._tri_call(self.log_mask, params, J, templ, outside, scratch,
                                None, stream)
            _lib.check(rc, 'rvs_template_tri')
        elif self.kind == 'nn':
            # scratch: dict(a0, a1 float32 [>=J, width], torch_stream) -- the
            # optimiser's rounds evaluate the MLP rows of all live simplices in
            # one rvs_template_nn call on the arm's side stream
            import ctypes
            nl = len(self.nn_W)
            Wp = (ctypes.c_void_p * nl)(*[w.data_ptr() for w in self.nn_W])
            bp = (ctypes.c_void_p * nl)(*[b.data_ptr() for b in self.nn_b])
            rc = L.rvs_template_nn(
                _lib.ptr(params), J, self.ndim, self.log_mask,
                _lib.ptr(self.nn_M), _lib.ptr(self.nn_S), nl,
                ctypes.cast(Wp, ctypes.c_void_p),
                ctypes.cast(bp, ctypes.c_void_p), _lib.ptr(self.nn_dims),
                _lib.ptr(scratch['a0']), _lib.ptr(scratch['a1']),
                _lib.ptr(templ), stream)
            _lib.check(rc, 'rvs_template_nn')
            self._nn_outside(params[:J], out=outside, stream=stream)
        else:
            raise NotImplementedError(self.kind)

    def nn_width(self):
        return int(max(self.nn_dims[:-1]))

    def _eval_nn(self, params, templ, outside, mapped=False):
        import ctypes
        L = _lib.lib()
        J = params.shape[0]
        nl = len(self.nn_W)
        width = int(max(self.nn_dims[:-1]))
        a0 = torch.empty((J, width), dtype=torch.float32, device=self.device)
        a1 = torch.empty((J, width), dtype=torch.float32, device=self.device)
        Wp = (ctypes.c_void_p * nl)(*[w.data_ptr() for w in self.nn_W])
        bp = (ctypes.c_void_p * nl)(*[b.data_ptr() for b in self.nn_b])
        from . import engine
        with engine._ktime('template_nn', J):
            if mapped:   # Mapper.forward already applied: identity here
                M = torch.zeros_like(self.nn_M)
                Sc = torch.ones_like(self.nn_S)
            else:
                M, Sc = self.nn_M, self.nn_S
            rc = L.rvs_template_nn(_lib.ptr(params), J, self.ndim,
                                   0 if mapped else self.log_mask,
                                   _lib.ptr(M), _lib.ptr(Sc), nl,
                                   ctypes.cast(Wp, ctypes.c_void_p),
                                   ctypes.cast(bp, ctypes.c_void_p),
                                   _lib.ptr(self.nn_dims), _lib.ptr(a0),
                                   _lib.ptr(a1), _lib.ptr(templ), _lib.stream())
        _lib.check(rc, 'rvs_template_nn')
        outside.copy_(self._nn_outside(params, mapped))
        return templ, outside

    def hull_device(self):
        """(xeqs, yeqs) facet equations of the two convex hulls on the device,
        or None for a library without an outside check"""
        if self.nn_hull is None:
            return None
        if getattr(self, '_hull_dev', None) is None:
            self._hull_dev = tuple(
                torch.as_tensor(np.ascontiguousarray(h, dtype=np.float64),
                                device=self.device) for h in self.nn_hull)
        return self._hull_dev

    def _nn_outside(self, params, mapped=False, out=None, stream=None):
        """OutsideInterpolator.__call__ (nn/RVSInterpolator.py:63-71) on the
        Mapper-transformed point, as SpecInterpolator.outsideFlag does
        (spec_inter.py:257-272): squared positive distance to the facets of two
        convex hulls (facet equations built once on the host, kept on the
        device; rvs_nn_outside, one launch).  out / stream: caller-owned result
        rows and HIP stream (the optimiser's rounds)."""
        J = params.shape[0]
        if out is None:
            out = torch.empty(J, dtype=torch.float64, device=self.device)
        if self.nn_hull is None:
            # on the caller's stream, like the launch below (the optimiser's
            # chain reads the flags there)
            rc = _lib.lib().rvs_nn_outside(
                None, J, self.ndim, 0, None, None, 1, None, 0, None, 0,
                _lib.ptr(out), _lib.stream() if stream is None else stream)
            _lib.check(rc, 'rvs_nn_outside')
            return out
        xe, ye = self.hull_device()
        if params.dtype != torch.float64:
            raise TypeError('rvs_nn_outside takes float64 parameters')
        p = params
        rc = _lib.lib().rvs_nn_outside(
            _lib.ptr(p.contiguous()), J, self.ndim, self.log_mask,
            _lib.ptr(self.nn_M), _lib.ptr(self.nn_S), 1 if mapped else 0,
            _lib.ptr(xe), xe.shape[0], _lib.ptr(ye), ye.shape[0], _lib.ptr(out),
            _lib.stream() if stream is None else stream)
        _lib.check(rc, 'rvs_nn_outside')
        return out
