"""Likelihood core -- API mirror of py/rvspecfit/spec_fit.py on the MI355X engine.

Same names, argument meaning and error behaviour as the reference for one
spectrum (a list of `SpecData`), plus batched use: every function that takes
`specdata` also accepts an `engine.SpecBatch` (S spectra x arms resident in
HBM); scalars then become [S] tensors and results are device tensors stacked
on a leading batch axis.
"""
import logging
import random
import collections
import threading

import numpy as np
import torch

from . import _lib
from . import engine
from . import spec_inter
from .engine import SpecBatch, SPEED_OF_LIGHT  # noqa: F401
import functools


class ResolMatrix:
    """spec_fit.ResolMatrix (spec_fit.py:54-67): a (banded, scipy.sparse)
    resolution matrix applied to the resampled template"""

    def __init__(self, mat):
        self.fd = {'mat': mat}
        self.objid = random.getrandbits(128)

    def __hash__(self):
        return self.objid

    @property
    def mat(self):
        return self.fd['mat']


def construct_resol_mat(lam, resol=None, width=None):
    """spec_fit.construct_resol_mat (spec_fit.py:410-471): Gaussian rows of
    sigma lam/resol/2.35 (or `width` Angstrom), truncated at 5 sigma, each row
    normalised; host code (built once per grid)."""
    import scipy.sparse
    assert (resol is None or width is None)
    assert (resol is not None or width is not None)
    lam = np.asarray(lam, dtype=np.float64)
    if resol is not None:
        sigs = lam / resol / 2.35
    elif np.isscalar(width):
        sigs = np.zeros(len(lam)) + width
    else:
        sigs = np.asarray(width, dtype=np.float64)
    thresh = 5
    assert (np.all(np.diff(lam) > 0))
    n = len(lam)
    i1 = np.maximum(np.searchsorted(lam, lam - thresh * sigs, 'left'), 0)
    i2 = np.minimum(np.searchsorted(lam, lam + thresh * sigs, 'right'), n - 1)
    pix = np.arange(n)
    maxl = min(n, max(np.max(i2 - pix), np.max(pix - i1)))
    offsets = np.arange(-maxl, maxl + 1)
    xs2d = pix[None, :] + offsets[:, None]
    mask = (xs2d >= 0) & (xs2d < n)
    xs2d[~mask] = 0
    XL = np.exp(-0.5 * ((lam[xs2d] - lam[None, :]) / sigs[None, :])**2) * mask
    XL = XL / XL.sum(axis=0)[None, :]
    yids = (pix[None, :] + (n - offsets)[:, None]) % n
    xids = yids * 0 + maxl + offsets[:, None]
    XL = XL[xids, yids]
    return ResolMatrix(scipy.sparse.spdiags(XL, offsets, n, n))


def convolve_resol(spec, resol_matrix):
    """spec_fit.convolve_resol (spec_fit.py:474-492); host API.  Inside the
    likelihood kernels the matrix is applied on the device (rvs_chisq_grid_resol
    and the `taps` argument of rvs_chisq_full / rvs_chisq_point)."""
    return resol_matrix.mat @ spec


class SpecData:
    """spec_fit.SpecData (spec_fit.py:70-145): one spectroscopic dataset."""

    def __init__(self, name, lam, spec, espec, badmask=None, resolution=None,
                 dtype=np.float64):
        fd = {}
        fd['name'] = name
        fd['lam'] = np.ascontiguousarray(lam, dtype=dtype)
        fd['spec'] = np.ascontiguousarray(spec, dtype=dtype)
        fd['espec'] = np.ascontiguousarray(espec, dtype=dtype)
        fd['resolution'] = resolution
        fd['spec_error_ratio'] = np.ascontiguousarray(fd['spec'] / fd['espec'],
                                                      dtype=dtype)
        if badmask is None:
            badmask = np.zeros(len(fd['spec']), dtype=bool)
        fd['badmask'] = np.asarray(badmask, dtype=bool)
        # (the reference freezes the record, utils.freezeDict: the fields have no
        # setters -- the device copy of a batch is cached under objid)
        self.fd = fd
        self.objid = random.getrandbits(128)

    name = property(lambda self: self.fd['name'])
    lam = property(lambda self: self.fd['lam'])
    spec = property(lambda self: self.fd['spec'])
    espec = property(lambda self: self.fd['espec'])
    spec_error_ratio = property(lambda self: self.fd['spec_error_ratio'])
    badmask = property(lambda self: self.fd['badmask'])
    resolution = property(lambda self: self.fd['resolution'])

    def __hash__(self):
        return self.objid


def get_poly_basis(lam, npoly, rbf=True):
    """spec_fit.get_poly_basis (spec_fit.py:148-176): the continuum basis on the
    wavelength grid `lam`, [npoly, len(lam)] float64 -- three monomials and npoly - 3
    Gaussians on the grid's range mapped to [-1, 1] (rbf), or Chebyshev polynomials.
    Built by the kernel the likelihood's own tables come from (rvs_basis_build,
    csrc/tables.hip); numpy in, numpy out."""
    _lib.require_gpu()
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    n = len(lam)
    dl = torch.as_tensor(lam[None, :]).to('cuda')
    raw = torch.empty((1, n + 1, npoly), dtype=torch.float64, device='cuda')
    cen = torch.as_tensor(np.linspace(-1, 1, max(npoly - 3, 1), True)).to('cuda')
    rc = _lib.lib().rvs_basis_build(_lib.ptr(dl), None, 1, n, npoly,
                                    int(bool(rbf)), _lib.ptr(cen), _lib.ptr(raw),
                                    None, None, _lib.stream())
    _lib.check(rc, 'rvs_basis_build')
    return np.ascontiguousarray(raw[0, :n].T.cpu().numpy())


@functools.lru_cache(100)
def get_basis(specdata, npoly, rbf=True):
    """spec_fit.get_basis (spec_fit.py:179-200): get_poly_basis of a SpecData's
    wavelengths, cached per dataset"""
    return get_poly_basis(specdata.lam, npoly, rbf=rbf)


def get_chisq0(spec, templ, polys, get_coeffs=False, espec=None):
    """spec_fit.get_chisq0 (spec_fit.py:306-354): -2 log L of `spec` given the
    template `templ` ON THE SAME PIXELS, marginalised over the linear continuum
    coefficients of the basis `polys` [npoly, npix]:
        log det(ST ST^T) + 2 sum log e + |D - a ST|^2,  ST = polys templ / e, D = spec / e
    (espec None: spec and templ are taken as already divided by the uncertainty).
    Cholesky, the eigen decomposition for a matrix that does not factor (the
    reference's SVD tier) -- rvs_chisq_full(unit_template=2), one block per row.

    One spectrum: 1-D numpy arrays in, a float (and the coefficients [npoly] with
    get_coeffs) out, as the reference.  Batched: spec / templ / espec [S, npix]
    (numpy or device tensors; templ may be [npix] = one template for all) give
    device tensors chisq [S] (and coeffs [S, npoly])."""
    _lib.require_gpu()
    single = np.ndim(spec) == 1

    def dev(x):
        if torch.is_tensor(x):
            return x.to('cuda', torch.float64)
        return torch.as_tensor(np.asarray(x, dtype=np.float64)).to('cuda')
    sp = dev(spec).reshape(-1, np.shape(spec)[-1]).contiguous()
    S, npix = sp.shape
    tp = dev(templ).reshape(-1, npix).contiguous()
    assert tp.shape[0] in (1, S)
    es = (torch.ones_like(sp) if espec is None
          else dev(espec).reshape(-1, npix).expand(S, npix).contiguous())
    pT = dev(polys).T.contiguous()
    npoly = pT.shape[1]
    assert pT.shape[0] == npix
    f64 = dict(dtype=torch.float64, device='cuda')
    i32 = dict(dtype=torch.int32, device='cuda')
    chisq = torch.empty(S, **f64)
    coeffs = torch.empty((S, npoly), **f64)
    tchi = torch.empty(S, **f64)
    ngood = torch.empty(S, **i32)
    status = torch.zeros(S, **i32)
    jt = (torch.zeros(S, **i32) if tp.shape[0] == 1
          else torch.arange(S, **i32))
    rc = _lib.lib().rvs_chisq_full(
        None, _lib.ptr(pT), _lib.ptr(sp), _lib.ptr(es), None, npix, npoly, S,
        None, _lib.ptr(tp), 0, tp.shape[0], 1, 1, 2, None, _lib.ptr(jt), S, None,
        0.0, 0, None, 0, 0, _lib.ptr(chisq), _lib.ptr(coeffs), None, None,
        _lib.ptr(tchi), _lib.ptr(ngood), _lib.ptr(status), _lib.stream())
    _lib.check(rc, 'rvs_chisq_full')
    if single:
        c = float(chisq[0].item())
        return (c, coeffs[0].cpu().numpy()) if get_coeffs else c
    return (chisq, coeffs) if get_coeffs else chisq


def compute_vsini_kernel(R, eps=0.6):
    """spec_fit.compute_vsini_kernel (spec_fit.py:565-625): the 2 ceil(R + 1) + 1
    weights of the rotational kernel of half width R pixels (linear limb darkening
    eps, integrated against a piecewise-linear signal), normalised -- the response
    of rvs_vsini_convolve to a unit pulse: every output is a sum of exact zeros and
    one weight times 1.0."""
    _lib.require_gpu()
    assert R > 0
    kmax = int(np.ceil(R + 1))
    n = 4 * kmax + 8   # (the kernel must fit the row: csrc/template.hip)
    c = n // 2
    x = torch.zeros((1, n), dtype=torch.float64, device='cuda')
    x[0, c] = 1.0
    out = torch.empty_like(x)
    lnstep = 1e-4
    vs = torch.as_tensor([R * lnstep * SPEED_OF_LIGHT], dtype=torch.float64).to('cuda')
    rc = _lib.lib().rvs_vsini_convolve(_lib.ptr(x), _lib.ptr(vs), None, lnstep,
                                       float(eps), n, 1, _lib.ptr(out),
                                       _lib.stream())
    _lib.check(rc, 'rvs_vsini_convolve')
    return out[0, c - kmax:c + kmax + 1].cpu().numpy()


class LRUDict:

    def __init__(self, N):
        self.N = N
        self.D = collections.OrderedDict()
        self.lock = threading.RLock()

    def __contains__(self, x):
        return x in self.D

    def __setitem__(self, x, y):
        if x not in self.D and len(self.D) == self.N:
            del self.D[next(iter(self.D))]
        self.D[x] = y
        self.D.move_to_end(x)

    def __getitem__(self, x):
        self.D.move_to_end(x)
        return self.D[x]

    def get_or_create(self, key, make):
        """cache[key], made by make() under the cache's lock if absent.  Device
        tensors made here are complete before another host thread (on another
        stream, vel_fit._process_split) can see the entry."""
        with self.lock:
            if key not in self.D:
                val = make()
                if torch.cuda.is_available():
                    torch.cuda.current_stream().synchronize()
                self[key] = val
            return self[key]


_batch_cache = LRUDict(16)


def as_batch(specdata):
    """list of SpecData (one spectrum) -> cached single-spectrum SpecBatch."""
    if isinstance(specdata, SpecBatch):
        return specdata, True
    if isinstance(specdata, SpecData):
        specdata = [specdata]
    key = tuple(sd.objid for sd in specdata)
    return _batch_cache.get_or_create(
        key, lambda: SpecBatch.from_specdata([list(specdata)])), False


def _params_tensor(atm_params, S, ndim, dev):
    if isinstance(atm_params, torch.Tensor):
        p = atm_params.to(dev, torch.float64)
    else:
        p = torch.as_tensor(np.asarray(atm_params, dtype=np.float64)).to(dev)
    if p.dim() == 1:
        p = p[None, :].expand(S, ndim)
    return p.contiguous()


def _vsini_tensor(rot_params, S, dev):
    if rot_params is None:
        return None
    if isinstance(rot_params, torch.Tensor):
        v = rot_params.to(dev, torch.float64)
    else:
        v = torch.as_tensor(np.asarray(rot_params, dtype=np.float64)).to(dev)
    v = v.reshape(-1)
    if v.numel() == 1:
        v = v.expand(S)
    return v.contiguous()


_resol_cache = LRUDict(16)


def _resols(batch, resol_params):
    """`resol_params` {setup: ResolMatrix} (shared by every spectrum of the arm,
    spec_fit.py:866-867, 922-923) -> per-arm device taps, or None"""
    if resol_params is None:
        return None
    out = []
    for arm in batch.arms:
        R = resol_params[arm.name]
        if arm.resol is not None:
            raise ValueError('You are not allowed to set resol_param together '
                             'with the resolution of each SpecData')
        if arm.G > 1:
            # (the reference applies resol_params[setup] to every spectrum of the
            # setup: one matrix cannot fit wavelength grids of different length)
            raise ValueError('resol_params needs one wavelength grid per setup; '
                             'give each SpecData its own resolution')
        key = (hash(R), arm.S, str(batch.device))
        def make(R=R, arm=arm):
            taps, nd = engine.resol_taps([R.mat], arm.npix)
            return engine.make_resol(taps, nd, arm.S, batch.device)
        out.append(_resol_cache.get_or_create(key, make))
    return out


def _overlap_check(templ_l0, templ_l1, spec_l0, spec_l1, min_vel, max_vel):
    # spec_fit.py:786-794
    for vel in [min_vel, max_vel]:
        corr = np.sqrt((1 + vel / SPEED_OF_LIGHT) / (1 - vel / SPEED_OF_LIGHT))
        if templ_l0 * corr > spec_l0 or templ_l1 * corr < spec_l1:
            raise RuntimeError(
                f"The template library ({templ_l0},{templ_l1})  doesn't cover"
                f" this wavelength range ({spec_l0},{spec_l1}) with "
                f"velocities {min_vel} {max_vel}")


def _check_overlap_all(batch, libs, config, vmin, vmax):
    for arm in batch.arms:
        lib = libs[arm.name]
        # (a grid set: the widest extent over the spectra's own grids)
        _overlap_check(lib.lam[0], lib.lam[-1], min(g[0] for g in arm.grids),
                       max(g[-1] for g in arm.grids),
                       min(config['min_vel'], vmin), max(config['max_vel'], vmax))


def _raise_for_status(st, what):
    """batch-of-1: turn status bits back into the reference's exceptions"""
    if st & _lib.ST_SPLINE_GRID:
        raise AssertionError('spline knots not uniformly spaced')  # spliner.py:51
    if st & _lib.ST_SPLINE_RANGE:
        raise AssertionError('spline evaluated outside its knots')
    if st & _lib.ST_NONFINITE:
        raise RuntimeError('The log(likelihood) value is not finite when '
                           'processing ' + what)


def getCurTempl(spec_setup, atm_param, rot_params, config):
    """spec_fit.getCurTempl (spec_fit.py:357-407), numpy outputs."""
    interp = spec_inter.getInterpolator(spec_setup, config)
    lib = interp.lib
    p = _params_tensor(tuple(atm_param), 1, lib.ndim, lib.device)
    vs = _vsini_tensor(rot_params, 1, lib.device)
    _, outside, templ = engine.build_templates(lib, p, vs, return_templ=True)
    return (float(outside[0].item()), lib.lam, templ[0].cpu().numpy(),
            random.getrandbits(128), lib.log_step)


def convolve_vsini(lam_templ, templ, vsini, eps=0.6):
    """spec_fit.convolve_vsini (spec_fit.py:628-682), numpy in / numpy out."""
    _lib.require_gpu()
    t = torch.as_tensor(np.ascontiguousarray(templ, dtype=np.float64)[None, :]
                        ).to('cuda')
    v = torch.as_tensor(np.array([vsini], dtype=np.float64)).to('cuda')
    return engine.convolve_vsini(np.asarray(lam_templ), t, v,
                                 eps)[0].cpu().numpy()


class _DeviceSpline:
    """spliner.Spline look-alike (spliner.py:8-53) on the device kernels."""

    def __init__(self, xs, ys, log_step=True):
        _lib.require_gpu()
        assert xs.dtype == np.float64 and ys.dtype == np.float64
        self.xs = np.ascontiguousarray(xs)
        self.N = len(xs)
        self.log_step = int(log_step)
        self.knots = torch.as_tensor(self.xs).to('cuda')
        y = torch.as_tensor(np.ascontiguousarray(ys)[None, :]).to('cuda')
        self.coef = torch.empty((1, self.N, 4), dtype=torch.float64,
                                device='cuda')
        # form 0: the reference's A, B, C, D (spliner.c:52-59)
        rc = _lib.lib().rvs_spline_construct(_lib.ptr(self.knots), _lib.ptr(y),
                                             self.N, 1, 0, None,
                                             _lib.ptr(self.coef), _lib.stream())
        _lib.check(rc, 'rvs_spline_construct')

    def __call__(self, evalx, return_pos=False):
        ex = torch.as_tensor(np.ascontiguousarray(evalx, dtype=np.float64)[None]
                             ).to('cuda')
        n = ex.shape[1]
        ret = torch.empty((1, n), dtype=torch.float64, device='cuda')
        pos = torch.empty((1, n), dtype=torch.int32, device='cuda')
        st = torch.zeros(1, dtype=torch.int32, device='cuda')
        rc = _lib.lib().rvs_spline_eval(_lib.ptr(self.knots),
                                        _lib.ptr(self.coef), self.N,
                                        self.log_step, _lib.ptr(ex), n, 1,
                                        _lib.ptr(ret), _lib.ptr(pos),
                                        _lib.ptr(st), _lib.stream())
        _lib.check(rc, 'rvs_spline_eval')
        assert int(st.item()) == 0  # spliner.py:51
        if return_pos:
            return ret[0].cpu().numpy(), pos[0].cpu().numpy()
        return ret[0].cpu().numpy()


def getRVInterpol(lam_templ, templ, log_step=True):
    return _DeviceSpline(lam_templ, templ, log_step=log_step)


def evalRV(interpol, vel, lams):
    beta = vel / SPEED_OF_LIGHT
    return interpol(lams * np.sqrt((1 - beta) / (1 + beta)))


def param_dict_to_tuple(paramDict, setup, config):
    interp = spec_inter.getInterpolator(setup, config)
    return tuple([paramDict[_] for _ in interp.parnames])


def get_chisq(specdata, vel, atm_params, rot_params=None, resol_params=None,
              options=None, config=None, cache=None, full_output=False,
              fast_interp=False, espec_systematic=None, outside_penalty=True):
    """spec_fit.get_chisq (spec_fit.py:797-989).

    One spectrum: returns a float (or the full_output dict of numpy arrays).
    SpecBatch: vel [S], atm_params [S, ndim] (or one tuple), rot_params None or
    vsini [S]; returns a device tensor [S] (or dict of tensors) and never
    raises for per-spectrum conditions (see 'status')."""
    options = options or {}
    npoly = options.get('npoly') or 5
    rbf = options.get('rbf_continuum', True)
    batch, is_batch = as_batch(specdata)
    S, dev = batch.S, batch.device
    resols = _resols(batch, resol_params)
    libs = spec_inter.get_libs(batch.names, config)
    ndim = libs[batch.names[0]].ndim
    params = _params_tensor(atm_params, S, ndim, dev)
    vsini = _vsini_tensor(rot_params, S, dev)
    if isinstance(vel, torch.Tensor):
        velt = vel.to(dev, torch.float64).reshape(S, 1)
    else:
        velt = torch.as_tensor(np.asarray(vel, dtype=np.float64)).to(dev)
        velt = velt.reshape(-1, 1).expand(S, 1).contiguous()
    vmin, vmax = float(velt.min().item()), float(velt.max().item())
    _check_overlap_all(batch, libs, config, vmin, vmax)
    # spec_fit.py:933-940: one value, or one per setup
    if isinstance(espec_systematic, dict):
        esys = [float(espec_systematic[n]) for n in batch.names]
    else:
        esys = float(espec_systematic) if espec_systematic is not None else 0.0
    coefs, outs = [], []
    for arm in batch.arms:
        c, o = engine.build_templates(libs[arm.name], params, vsini)
        coefs.append(c)
        outs.append(o)
    chisq, status = engine.chisq_point(batch, libs, coefs, outs, velt[:, 0],
                                       npoly=npoly, rbf=rbf, espec_sys=esys,
                                       outside_penalty=outside_penalty,
                                       resols=resols, fast_interp=fast_interp)
    # reference: a non finite arm value of an OUTSIDE template is skipped with
    # a warning instead of raising (spec_fit.py:963-969); we flag it in status
    if not is_batch:
        st = int(status[0].item())
        anyout = any(float(o[0].item()) > 0 for o in outs)
        if not (st & _lib.ST_NONFINITE and anyout):
            _raise_for_status(st, f'velocity {vel}, atm parameters {atm_params}')
    if not full_output:
        return chisq if is_batch else float(chisq[0].item())
    full = engine.chisq_full(batch, libs, coefs, velt[:, 0].contiguous(),
                             npoly=npoly, rbf=rbf, espec_sys=esys, resols=resols,
                             fast_interp=fast_interp)
    ret = {}
    ret['chisq'] = chisq if is_batch else float(chisq[0].item())
    ret['logl'] = -0.5 * ret['chisq']
    nanarm = [~torch.isfinite(o) for o in outs]
    ca = [torch.where(n, torch.full_like(f['true_chisq'], float('nan')),
                      f['true_chisq']) for f, n in zip(full, nanarm)]
    na = [f['ngood'] for f in full]
    if is_batch:
        ret['chisq_array'] = torch.stack(ca, dim=1)
        ret['npix_array'] = torch.stack(na, dim=1)
        ret['red_chisq_array'] = ret['chisq_array'] / ret['npix_array']
        ret['models'] = [f['model'] for f in full]
        ret['raw_models'] = [f['raw_model'] for f in full]
        ret['status'] = status
    else:
        ret['chisq_array'] = [float(c[0].item()) for c in ca]
        ret['npix_array'] = [int(n[0].item()) for n in na]
        ret['red_chisq_array'] = [c / n for c, n in zip(ret['chisq_array'],
                                                        ret['npix_array'])]
        ret['models'] = [f['model'][0].cpu().numpy() for f in full]
        ret['raw_models'] = [f['raw_model'][0].cpu().numpy() for f in full]
    return ret


# rows per launch set of the from-template objective (template buffers of
# 3 arms x 32 768 rows x ~6000 px x 8 B = 4.7 GB)
FROM_TEMPLATE_CHUNK = 32768


def chisq_jobs(batch, idx, vel, params, vsini, options, config,
               outside_penalty=True, espec_systematic=None, resol_params=None):
    """get_chisq for J jobs: job j is spectrum idx[j] against its own template
    (params[j], vsini[j]) at velocity vel[j] (rvs_chisq_point: one lane per
    job, residual norm formed explicitly as in spec_fit.py:249).
    Returns chisq [J], status [J]."""
    options = options or {}
    npoly = options.get('npoly') or 5
    rbf = options.get('rbf_continuum', True)
    libs = spec_inter.get_libs(batch.names, config)
    params = params.contiguous()
    esys = float(espec_systematic) if espec_systematic is not None else 0.0
    resols = _resols(batch, resol_params)
    js = idx.to(torch.int32).contiguous()
    if engine.can_fuse_objective(batch, libs, resols, npoly=npoly):
        return engine.objective_fused(batch, libs, params, vsini, vel,
                                      npoly=npoly, rbf=rbf, job_spec=js,
                                      espec_sys=esys,
                                      outside_penalty=outside_penalty)
    if engine.can_fuse_objective(batch, libs, resols, npoly=npoly,
                                 from_template=True):
        # MLP / Delaunay evaluators: template rows from their own kernel, then
        # broadening + spline + chi^2 in one kernel.  The rows ([J, ntp] float64
        # per arm) go through HBM: in chunks, so that the Hessian stage's 33+
        # evaluations per spectrum of a 10 000-spectra batch reuse three 1.6 GB
        # buffers instead of allocating 49 GB (1.9 -> 0.5 s of that stage)
        J = int(js.shape[0])
        out = torch.empty(J, dtype=torch.float64, device=batch.device)
        st = torch.empty(J, dtype=torch.int32, device=batch.device)
        vel = vel.to(device=batch.device, dtype=torch.float64)
        for a in range(0, J, FROM_TEMPLATE_CHUNK):
            b = min(J, a + FROM_TEMPLATE_CHUNK)
            tt = [libs[arm.name].eval_batch(params[a:b]) for arm in batch.arms]
            out[a:b], st[a:b] = engine.objective_from_template(
                batch, libs, [t[0] for t in tt], [t[1] for t in tt],
                None if vsini is None else vsini[a:b], vel[a:b], npoly=npoly,
                rbf=rbf, job_spec=js[a:b], espec_sys=esys,
                outside_penalty=outside_penalty)
            del tt
        return out, st
    coefs, outs = [], []
    for arm in batch.arms:
        c, o = engine.build_templates(libs[arm.name], params, vsini)
        coefs.append(c)
        outs.append(o)
    return engine.chisq_point(batch, libs, coefs, outs, vel, npoly=npoly,
                              rbf=rbf, job_spec=js, espec_sys=esys,
                              outside_penalty=outside_penalty, resols=resols)


def chisq_grid_jobs(batch, vel_grid, params, vsini, options, config,
                    outside_penalty=True, espec_systematic=None,
                    resol_params=None, spec_idx=None, shared_vsini=None):
    """chi^2 [S, Np, Nv] for params [S, Np, ndim] (device) on a shared or
    per-spectrum velocity grid: the double loop of find_best as one launch set.
    spec_idx (device int [S']): rows refer to these spectra of the batch (the
    batch's per-spectrum preparation is reused, nothing is copied).
    params [Np, ndim] (2-D): ONE parameter list for every spectrum, with
    `shared_vsini` (float or None) the rotation of all of them -- the Np
    templates are then built once and every spectrum's jobs point at them, as
    the reference's template caches serve find_best's double loop
    (spec_fit.py:1043-1060, 357-407); same values as Np templates per spectrum."""
    options = options or {}
    npoly = options.get('npoly') or 5
    rbf = options.get('rbf_continuum', True)
    dev = batch.device
    S = batch.S if spec_idx is None else int(spec_idx.shape[0])
    libs = spec_inter.get_libs(batch.names, config)
    one_list = params.dim() == 2
    Np = params.shape[0] if one_list else params.shape[1]
    if one_list:
        flat = params.contiguous()
        vs = None if shared_vsini is None else torch.full(
            (Np, ), float(shared_vsini), dtype=torch.float64, device=dev)
    else:
        flat = params.reshape(S * Np, -1).contiguous()
        vs = None
        if vsini is not None:
            vs = vsini.reshape(S, -1).expand(S, Np).reshape(-1).contiguous()
    coefs, outs = [], []
    for arm in batch.arms:
        c, o = engine.build_templates(libs[arm.name], flat, vs)
        coefs.append(c)
        outs.append(o)
    rows = torch.arange(S, dtype=torch.int32, device=dev) \
        if spec_idx is None else spec_idx.to(torch.int32)
    job_spec = rows.repeat_interleave(Np).contiguous()
    job_templ = None
    if one_list:
        job_templ = torch.arange(Np, dtype=torch.int32,
                                 device=dev).repeat(S).contiguous()
    vg = vel_grid
    if vg.dim() == 2:  # per spectrum grids -> per job
        vg = vg.repeat_interleave(Np, dim=0).contiguous()
    esys = float(espec_systematic) if espec_systematic is not None else 0.0
    chisq, status = engine.chisq_grid(batch, libs, coefs, outs, vg,
                                      npoly=npoly, rbf=rbf, job_spec=job_spec,
                                      job_templ=job_templ, espec_sys=esys,
                                      outside_penalty=outside_penalty,
                                      resols=_resols(batch, resol_params))
    if one_list:   # callers index the outside flags per (spectrum, parameter)
        outs = [o.repeat(S) for o in outs]
    return chisq.reshape(S, Np, -1), status.reshape(S, Np), outs


def find_best(specdata, vel_grid, params_list, rot_params=None,
              resol_params=None, options=None, config=None, quadratic=True):
    """spec_fit.find_best (spec_fit.py:1018-1092).

    One spectrum: same dict as the reference (numpy / floats).
    SpecBatch: params_list is [Np][ndim] (shared) or a tensor [S, Np, ndim];
    values are device tensors with a leading S axis."""
    batch, is_batch = as_batch(specdata)
    S, dev = batch.S, batch.device
    if isinstance(params_list, torch.Tensor):
        params = params_list.to(dev, torch.float64)
    else:
        params = torch.as_tensor(np.asarray(params_list, dtype=np.float64)).to(dev)
    # one list for every spectrum (the reference's call) and one rotation: the
    # templates are built once for the batch
    one_list = params.dim() == 2 and (
        rot_params is None or (not isinstance(rot_params, torch.Tensor)
                               and np.size(rot_params) == 1))
    plist = params if one_list else None
    if params.dim() == 2:
        params = params[None].expand(S, *params.shape)
    Np = params.shape[1]
    if isinstance(vel_grid, torch.Tensor):
        vg = vel_grid.to(dev, torch.float64)
    else:
        vg = torch.as_tensor(np.asarray(vel_grid, dtype=np.float64)).to(dev)
    libs = spec_inter.get_libs(batch.names, config)
    _check_overlap_all(batch, libs, config, float(vg.min().item()),
                       float(vg.max().item()))
    vsini = _vsini_tensor(rot_params, S, dev)
    if one_list:
        sv = None if rot_params is None else float(np.ravel(rot_params)[0])
        chisq, status, outs = chisq_grid_jobs(batch, vg, plist, None, options,
                                              config, resol_params=resol_params,
                                              shared_vsini=sv)
    else:
        chisq, status, outs = chisq_grid_jobs(batch, vg, params.contiguous(),
                                              vsini, options, config,
                                              resol_params=resol_params)
    res, probs, mst = engine.grid_moments(chisq.reshape(S * Np, -1), vg, Np=Np,
                                          quadratic=quadratic)
    i2 = res[:, 6].long()
    best_param = params[torch.arange(S, device=dev), i2]
    if is_batch:
        return dict(best_chi=res[:, 0], best_vel=res[:, 1], vel_err=res[:, 2],
                    kurtosis=res[:, 3], skewness=res[:, 4], best_param=best_param,
                    probs=probs, chisq=chisq, status=status, i1=res[:, 5].long(),
                    i2=i2, moment_status=mst)
    st = 0
    for v in status[0].tolist():
        st |= int(v)
    anyout = any(bool((o > 0).any().item()) for o in outs)
    if not (st & _lib.ST_NONFINITE and anyout):
        _raise_for_status(st, 'find_best')
    if int(mst[0].item()) & _lib.ST_QUAD_ASSERT:
        raise AssertionError('quadratic interpolation left its bracket')
    r = res[0].cpu().numpy()
    kur, skw = float(r[3]), float(r[4])
    if r[2] < 1e-10:
        kur, skw = 0, 0
    pl = params_list
    best = pl[int(r[6])] if not isinstance(pl, torch.Tensor) else \
        best_param[0].cpu().numpy()
    return dict(best_chi=float(r[0]), best_vel=float(r[1]), vel_err=float(r[2]),
                best_param=best, kurtosis=kur, skewness=skw,
                probs=probs[0].cpu().numpy())


def get_chisq_continuum(specdata, options=None):
    """spec_fit.get_chisq_continuum (spec_fit.py:739-783)."""
    options = options or {}
    npoly = options.get('npoly') or 5
    rbf = options.get('rbf_continuum', True)
    batch, is_batch = as_batch(specdata)
    full = engine.chisq_continuum_fix(
        batch, engine.chisq_continuum(batch, npoly=npoly, rbf=rbf), npoly=npoly,
        rbf=rbf)
    ca = torch.stack([f['true_chisq'] for f in full], dim=1)
    ng = torch.stack([f['ngood'] for f in full], dim=1)
    if is_batch:
        return dict(chisq_array=ca, redchisq_array=ca / ng)
    ca, ng = ca[0].cpu().numpy(), ng[0].cpu().numpy()
    return dict(chisq_array=ca, redchisq_array=ca / ng)
