"""CPU restatement (numpy/scipy + oracle_core.c) of the rvspecfit likelihood hot path.

TEST INFRASTRUCTURE ONLY.  This module is the *checker*: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  The
product package (rvspecfit_amd/) never does; it fails loudly without its HIP
library.

Parity status: PINNED.  Every function below is checked in
tests/test_oracle_golden.py against vectors captured by importing the reference
itself in the build container (tests/golden/make_golden.py -> cases.npz,
lib_*.npz), and the spline additionally against the reference's own C file
compiled in place (oracle/_ref/libspliner_ref.so) and against
scipy.interpolate.CubicSpline(bc_type='natural') as the reference's own
tests/test_spline.py does.

Third-party arithmetic kept at the same boundary as the reference (SURVEY
8(c) C2): numpy pocketfft (rfft/irfft), scipy.signal.medfilt,
scipy.stats.binned_statistic, scipy.interpolate.UnivariateSpline and
scipy.optimize.least_squares(loss='soft_l1').

All file:line citations are relative to /root/reference/py/rvspecfit/.
"""
import ctypes
import itertools
import math
import os

import numpy as np
import scipy.interpolate
import scipy.optimize
import scipy.signal
import scipy.stats

SPEED_OF_LIGHT = 299792.458  # spec_fit.py:23 (scipy.constants.speed_of_light/1e3)

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


def _load(path):
    if not os.path.exists(path):
        raise RuntimeError('%s missing: run `make -C oracle`' % path)
    return ctypes.CDLL(path)


_core = None


def core():
    global _core
    if _core is None:
        L = _load(_HERE + '/_build/liboracle_core.so')
        L.orc_spline_construct.argtypes = [_dp, _dp, ctypes.c_int] + [_dp] * 5
        L.orc_spline_construct.restype = None
        L.orc_spline_eval.argtypes = ([_dp, ctypes.c_int, ctypes.c_int] +
                                      [_dp] * 6 + [ctypes.c_int, _dp, _ip])
        L.orc_spline_eval.restype = ctypes.c_int
        L.orc_chisq0.argtypes = [_dp, _dp, _dp, _dp, ctypes.c_int, ctypes.c_int,
                                 _dp, _ip]
        L.orc_chisq0.restype = ctypes.c_double
        L.orc_chisq_vel.argtypes = ([_dp] * 4 + [ctypes.c_int, ctypes.c_int] +
                                    [_dp] * 6 + [ctypes.c_int, ctypes.c_int, _dp,
                                                 ctypes.c_int, _dp])
        L.orc_chisq_vel.restype = ctypes.c_int
        _core = L
    return _core


def _p(a):
    return a.ctypes.data_as(_dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# --------------------------------------------------------------------------
# A7  spline   (src/spliner.c:7-108, spliner.py:8-53)
# --------------------------------------------------------------------------
class Spline:

    def __init__(self, xs, ys, log_step=True, lib=None):
        xs, ys = _c(xs), _c(ys)
        N = len(xs)
        self.N, self.xs, self.log_step = N, xs, int(log_step)
        self.A, self.B, self.C, self.D, self.h = (np.zeros(N - 1)
                                                  for _ in range(5))
        self._lib = lib
        if lib is None:
            core().orc_spline_construct(_p(xs), _p(ys), N, _p(self.A),
                                        _p(self.B), _p(self.C), _p(self.D),
                                        _p(self.h))
        else:  # the reference's own C (oracle/_ref), same signature
            lib.construct(_p(xs), _p(ys), N, _p(self.A), _p(self.B),
                          _p(self.C), _p(self.D), _p(self.h))

    def __call__(self, evalx, return_pos=False):
        evalx = _c(evalx)
        n = len(evalx)
        ret = np.zeros(n)
        pos = np.zeros(n, dtype=np.int32)
        if self._lib is None:
            st = core().orc_spline_eval(_p(evalx), n, self.N, _p(self.xs),
                                        _p(self.h), _p(self.A), _p(self.B),
                                        _p(self.C), _p(self.D), self.log_step,
                                        _p(ret), pos.ctypes.data_as(_ip))
        else:
            st = self._lib.evaler(_p(evalx), n, self.N, _p(self.xs), _p(self.h),
                                  _p(self.A), _p(self.B), _p(self.C),
                                  _p(self.D), self.log_step, _p(ret))
        assert st == 0, st  # spliner.py:51
        if return_pos:
            return ret, pos
        return ret


def load_reference_spliner():
    """The reference's spliner.c built in place by oracle/Makefile (or None)."""
    path = _HERE + '/_ref/libspliner_ref.so'
    if not os.path.exists(path):
        return None
    L = ctypes.CDLL(path)
    L.construct.argtypes = [_dp, _dp, ctypes.c_int] + [_dp] * 5
    L.construct.restype = None
    L.evaler.argtypes = ([_dp, ctypes.c_int, ctypes.c_int] + [_dp] * 6 +
                         [ctypes.c_int, _dp])
    L.evaler.restype = ctypes.c_int
    return L


# --------------------------------------------------------------------------
# A1  SpecData   (spec_fit.py:70-145)
# --------------------------------------------------------------------------
class SpecData:

    def __init__(self, name, lam, spec, espec, badmask=None, resolution=None):
        # resolution: scipy.sparse matrix (ResolMatrix.mat, spec_fit.py:54-67)
        self.resolution = resolution
        self.name = name
        self.lam, self.spec, self.espec = _c(lam), _c(spec), _c(espec)
        if badmask is None:
            badmask = np.zeros(len(self.spec), dtype=bool)
        self.badmask = np.asarray(badmask, dtype=bool)


# --------------------------------------------------------------------------
# A2  continuum bases   (spec_fit.py:148-176)
# --------------------------------------------------------------------------
def get_poly_basis(lam, npoly, rbf=True):
    x = (lam - lam[0]) / (lam[-1] - lam[0]) * 2 - 1
    out = np.zeros((npoly, len(lam)))
    if rbf:
        nmono = 3
        for i in range(min(nmono, npoly)):
            out[i] = x**i
        nrbf = npoly - nmono
        if nrbf > 0:
            sig = 1. / nrbf
            cen = np.linspace(-1, 1, nrbf, True)
            out[nmono:] = np.exp(-0.5 * (x[None, :] - cen[:, None])**2 / sig**2)
    else:
        eye = np.eye(npoly)
        for i in range(npoly):
            out[i] = np.polynomial.Chebyshev(eye[i])(x)
    return out


# --------------------------------------------------------------------------
# A10  continuum marginalisation   (spec_fit.py:203-354)
# --------------------------------------------------------------------------
def get_chisq0(spec, templ, polys, get_coeffs=False, espec=None):
    """numpy form following _get_chisq0_svd (spec_fit.py:255-303); the C form
    orc_chisq0 is the Cholesky statement of the same quantity."""
    if espec is None:   # already divided by the uncertainty (spec_fit.py:260-263)
        D, nt, logz = spec, templ, 0.
    else:
        D = spec / espec
        nt = templ / espec
        logz = np.log(espec).sum()
    ST = nt[None, :] * polys
    v = ST @ D
    Minv = ST @ ST.T
    u, s, vt = np.linalg.svd(Minv)
    ldet = np.sum(np.log(s))
    a = vt.T @ ((1. / s)[:, None] * u.T) @ v
    chisq = ldet + 2 * logz + np.linalg.norm(D - a @ ST)**2
    if get_coeffs:
        return chisq, a
    return chisq


def get_chisq0_c(spec, templ, polys, espec, get_coeffs=False):
    polys = _c(polys)
    p, n = polys.shape
    co = np.zeros(p)
    st = ctypes.c_int(0)
    val = core().orc_chisq0(_p(_c(spec)), _p(_c(templ)), _p(polys),
                            _p(_c(espec)), p, n, _p(co), ctypes.byref(st))
    if get_coeffs:
        return val, co
    return val


# --------------------------------------------------------------------------
# A3  polylinear template interpolation
#     (spec_inter.py:62-194, read_grid.py:127-145)
# --------------------------------------------------------------------------
def _ccf_sets(d):
    """the CCF template sets of a converted artefact: keys ccf_* = continuum-
    normalised, ccfnc_* = rvs_make_ccf --nocontinuum (make_ccf.py:19-36)"""
    out = {}
    for cont, pre in ((True, 'ccf_'), (False, 'ccfnc_')):
        if pre + 'fft' not in d:
            continue
        out[cont] = dict(
            fft=d[pre + 'fft'], fft2=d[pre + 'fft2'], mod=d[pre + 'mod'],
            params=_c(d[pre + 'params']), vsinis=_c(d[pre + 'vsinis']),
            logl0=float(d[pre + 'logl0']), logl1=float(d[pre + 'logl1']),
            npoints=int(d[pre + 'npoints']),
            continuum=bool(d[pre + 'continuum']),
            splinestep=float(d[pre + 'splinestep'])
            if pre + 'splinestep' in d else None,
            maxcontpts=int(d[pre + 'maxcontpts']))
    return out


def _ccf_pick(lib, config):
    """get_ccf_info's choice (fitter_ccf.py:40-47): config key
    ccf_continuum_normalize, None / missing = True"""
    cont = (config or {}).get('ccf_continuum_normalize')
    cont = True if cont is None else bool(cont)
    if cont not in lib.ccf_sets:
        raise RuntimeError('no such CCF template set')
    return lib.ccf_sets[cont]


class Library:
    """One spectral setup loaded from the converted-artefact npz
    (same keys as tests/golden/lib_*.npz)."""

    def __init__(self, npz):
        d = dict(npz)
        self.lam = _c(d['lam'])
        self.dats = np.asarray(d['dats'])
        self.vec = _c(d['vec'])
        self.idgrid = np.asarray(d['idgrid'], dtype=np.int64)
        self.uvecs = [_c(d['uvec%d' % i]) for i in range(self.idgrid.ndim)]
        self.log_step = bool(d['log_step'])
        self.log_ids = [int(_) for _ in np.atleast_1d(d['log_ids'])]
        self.parnames = tuple(str(_) for _ in d['parnames'])
        self.ndim = len(self.uvecs)
        self.lens = np.array([len(_) for _ in self.uvecs])
        self.edges = np.array(list(itertools.product(*[[0, 1]] * self.ndim)))
        self.ptp = np.ptp(self.vec, axis=1)
        self.scaled = self.vec.T / self.ptp[None, :]
        self.exp = True  # log_spec default (spec_inter.py:331)
        self.ccf_sets = _ccf_sets(d)
        self.ccf = self.ccf_sets.get(True)

    def ccf_set(self, config):
        return _ccf_pick(self, config)

    # read_grid.py:127-145
    def map_params(self, p):
        v = np.array(p, dtype=np.float64)
        with np.errstate(all='ignore'):
            for i in self.log_ids:
                v[i] = np.log10(v[i])
        return v

    def nearest(self, mp):
        """brute-force statement of cKDTree.query on ptp-scaled coordinates
        (spec_inter.py:127-132); returns (distance, index)."""
        d2 = ((self.scaled - (mp / self.ptp)[None, :])**2).sum(axis=1)
        i = int(np.argmin(d2))
        return math.sqrt(d2[i]), i

    def cell(self, mp):
        return np.array([
            np.searchsorted(self.uvecs[i], mp[i], 'right') - 1
            for i in range(self.ndim)
        ])

    # spec_inter.py:77-92
    def outside_flag(self, p):
        mp = self.map_params(p)
        pos = self.cell(mp)
        outside = bool(np.any((pos < 0) | (pos >= self.lens - 1)))
        if not outside:
            ids = self.idgrid[tuple((pos[None, :] + self.edges).T)]
            outside = bool((ids == -1).any())
        if outside:
            if not np.isfinite(mp).all():
                return np.inf  # cKDTree.query of a non-finite point
            return self.nearest(mp)[0]
        return 0.

    # spec_inter.py:134-194
    def eval(self, p, details=False):
        mp = self.map_params(p)
        pos = self.cell(mp)
        info = dict(pos=pos, ids=None, weights=None, nearest=-1)
        FF = np.exp if self.exp else (lambda x: x)
        use_nn = bool(np.any((pos < 0) | (pos >= self.lens - 1)))
        if not use_nn:
            ids = self.idgrid[tuple((pos[None, :] + self.edges).T)]
            use_nn = bool(np.any(ids < 0))
        if use_nn:
            j = self.nearest(mp)[1] if np.isfinite(mp).all() else 0
            info['nearest'] = j
            spec = FF(self.dats[j].astype(np.float32))
        else:
            x = np.array([(mp[i] - self.uvecs[i][pos[i]]) /
                          (self.uvecs[i][pos[i] + 1] - self.uvecs[i][pos[i]])
                          for i in range(self.ndim)])
            w = np.prod(x[None, :]**self.edges * (1 - x[None, :])**(1 - self.edges),
                        axis=1)
            info['ids'], info['weights'] = ids, w
            spec = FF(np.dot(w, self.dats[ids, :]))
        spec = np.ascontiguousarray(spec, dtype=np.float64)
        if details:
            return spec, info
        return spec


class TriLibrary:
    """`interpolation_type = 'triangulation'` setup: spec_inter.TriInterp
    (spec_inter.py:11-59) on the exported arrays of the reference's Delaunay
    object; find_simplex restated as scipy's exhaustive search
    (_find_simplex_bruteforce: all barycentric coordinates in [-eps, 1+eps])."""

    def __init__(self, npz):
        d = dict(npz)
        self.lam = _c(d['lam'])
        self.dats = np.asarray(d['dats'], dtype=np.float64)
        self.simplices = np.asarray(d['simplices'])
        self.transform = np.asarray(d['transform'], dtype=np.float64)
        self.extraflags = np.asarray(d['extraflags'], dtype=np.float64)
        self.log_step = bool(d['log_step'])
        self.log_ids = [int(_) for _ in np.atleast_1d(d['log_ids'])]
        self.parnames = tuple(str(_) for _ in d['parnames'])
        self.ndim = self.transform.shape[2]
        self.exp = True
        # the CCF part of the setup is that of Library (make_ccf.py builds it from
        # whatever evaluator the setup has)
        self.ccf_sets = _ccf_sets(d)
        self.ccf = self.ccf_sets.get(True)

    map_params = Library.map_params

    def find_simplex(self, mp):
        if not np.isfinite(mp).all():
            return -1
        nd = self.ndim
        eps = 100 * np.finfo(float).eps
        c = np.einsum('sij,sj->si', self.transform[:, :nd, :],
                      mp[None, :] - self.transform[:, nd, :])
        cl = 1 - c.sum(axis=1)
        ok = np.all((c >= -eps) & (c <= 1 + eps), axis=1) & (cl >= -eps) & \
            (cl <= 1 + eps)
        w = np.nonzero(ok)[0]
        return int(w[0]) if len(w) else -1

    def _bary(self, mp, xid):
        nd = self.ndim
        b = np.empty(nd + 1)
        b[:nd] = self.transform[xid, :nd, :].dot(mp - self.transform[xid, nd, :])
        b[nd] = 1 - b[:nd].sum()
        return b

    def eval(self, p, details=False):
        mp = self.map_params(p)
        xid = self.find_simplex(mp)
        if xid == -1:
            spec = np.full(len(self.lam), np.nan)
            return (spec, dict(simplex=-1)) if details else spec
        b = self._bary(mp, xid)
        spec = (self.dats[self.simplices[xid], :] * b[:, None]).sum(axis=0)
        spec = np.exp(spec) if self.exp else spec
        if details:
            return spec, dict(simplex=xid, weights=b)
        return np.ascontiguousarray(spec)

    def outside_flag(self, p):
        mp = self.map_params(p)
        xid = self.find_simplex(mp)
        if xid == -1:
            return np.nan
        b = self._bary(mp, xid)
        return float((self.extraflags[self.simplices[xid]] * b).sum())


# --------------------------------------------------------------------------
# A6  rotational broadening   (spec_fit.py:495-682)
# --------------------------------------------------------------------------
def _rot_primitives(x, eps):
    x = np.clip(x, -1.0, 1.0)
    norm = np.pi * (1 - eps / 3.0)
    c1 = 2 * (1 - eps) / norm
    c2 = (np.pi / 2.0) * eps / norm
    s = np.sqrt(1 - x**2)
    k0 = c1 * (0.5 * (x * s + np.arcsin(x))) + c2 * (x - x**3 / 3.0)
    k1 = c1 * (-1.0 / 3.0 * (1 - x**2) * s) + c2 * (x**2 / 2.0 - x**4 / 4.0)
    return k0, k1


def _rot_segment(xa, xb, slope, intercept, eps):
    k0b, k1b = _rot_primitives(xb, eps)
    k0a, k1a = _rot_primitives(xa, eps)
    return slope * (k1b - k1a) + intercept * (k0b - k0a)


def compute_vsini_kernel(R, eps=0.6):
    assert R > 0
    kmax = int(np.ceil(R + 1))
    k = np.arange(0, kmax + 1)
    w = np.zeros(len(k))
    lo, hi = np.clip(k / R, -1, 1), np.clip((k + 1) / R, -1, 1)
    m = hi > lo
    if m.any():
        w[m] += _rot_segment(lo[m], hi[m], -R, 1 + k[m], eps)
    lo, hi = np.clip((k - 1) / R, -1, 1), np.clip(k / R, -1, 1)
    m = hi > lo
    if m.any():
        w[m] += _rot_segment(lo[m], hi[m], R, 1 - k[m], eps)
    full = np.concatenate([w[:0:-1], w])
    return full / full.sum()


def convolve_vsini(lam_templ, templ, vsini, eps=0.6):
    if vsini <= 0:
        return templ.copy()
    ratios = lam_templ[1:] / lam_templ[:-1]
    assert np.allclose(ratios, ratios[0])
    R = (vsini / SPEED_OF_LIGHT) / np.log(ratios[0])
    if R < 1e-9:
        return templ.copy()
    ker = compute_vsini_kernel(R, eps)
    # scipy.signal.convolve(mode='same') == zero-padded, centred direct sum
    return np.convolve(templ, ker, mode='same')


# --------------------------------------------------------------------------
# A5  getCurTempl   (spec_fit.py:357-407)
# --------------------------------------------------------------------------
def get_cur_templ(lib, atm_params, rot_params):
    outside = float(lib.outside_flag(atm_params))
    spec = lib.eval(atm_params)
    if outside > 0:
        mx = np.abs(spec).max()
        if mx > 1e100 or not np.isfinite(mx):
            outside = np.nan
    if np.isfinite(outside) and rot_params is not None:
        spec = convolve_vsini(lib.lam, spec, *rot_params)
    return outside, spec


# --------------------------------------------------------------------------
# A8/A11  get_chisq   (spec_fit.py:786-989)
# --------------------------------------------------------------------------
def _overlap_check(t0, t1, s0, s1, min_vel, max_vel):
    for vel in (min_vel, max_vel):
        corr = np.sqrt((1 + vel / SPEED_OF_LIGHT) / (1 - vel / SPEED_OF_LIGHT))
        if t0 * corr > s0 or t1 * corr < s1:
            raise RuntimeError('template does not cover the data')


def eval_rv(spl, vel, lam):  # spec_fit.py:707-727
    beta = vel / SPEED_OF_LIGHT
    return spl(lam * np.sqrt((1 - beta) / (1 + beta)))


def construct_resol_mat(lam, resol=None, width=None):
    """A9: spec_fit.construct_resol_mat (spec_fit.py:410-471) -> scipy.sparse"""
    import scipy.sparse
    lam = np.asarray(lam, dtype=float)
    if resol is not None:
        sigs = lam / resol / 2.35
    else:
        sigs = np.zeros(len(lam)) + width
    n = len(lam)
    i1 = np.maximum(np.searchsorted(lam, lam - 5 * sigs, 'left'), 0)
    i2 = np.minimum(np.searchsorted(lam, lam + 5 * sigs, 'right'), n - 1)
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
    return scipy.sparse.spdiags(XL[xids, yids], offsets, n, n)


def get_chisq(specdata, vel, atm_params, rot_params=None, options=None,
              config=None, libs=None, cache=None, full_output=False,
              espec_systematic=None, outside_penalty=True, use_c=False,
              resol_params=None):
    npoly = options.get('npoly') or 5
    rbf = options.get('rbf_continuum', True)
    accum = 0
    badchi = 10 * sum(len(_.lam) for _ in specdata)
    out = dict(chisq_array=[], red_chisq_array=[], npix_array=[], models=[],
               raw_models=[])
    for sd in specdata:
        lib = libs[sd.name]
        key = (sd.name, tuple(atm_params),
               None if rot_params is None else tuple(rot_params))
        if cache is not None and key in cache:
            outside, tspec, spl = cache[key]
        else:
            outside, tspec = get_cur_templ(lib, atm_params, rot_params)
            spl = None
        if not np.isfinite(outside):
            accum += 1000 * badchi
            out['chisq_array'].append(np.nan)
            out['red_chisq_array'].append(np.nan)
            out['models'].append(np.zeros(len(sd.lam)) + np.nan)
            continue
        if outside_penalty:
            accum += outside * badchi
        _overlap_check(lib.lam[0], lib.lam[-1], sd.lam[0], sd.lam[-1],
                       min(config['min_vel'], vel), max(config['max_vel'], vel))
        if spl is None:
            spl = Spline(lib.lam, tspec, log_step=lib.log_step)
            if cache is not None:
                cache[key] = (outside, tspec, spl)
        ev = eval_rv(spl, vel, sd.lam)
        # A9: resolution matrix (spec_fit.py:920-929)
        if resol_params is not None:
            ev = resol_params[sd.name] @ ev
        if sd.resolution is not None:
            if resol_params is not None:
                raise ValueError('resol_params together with SpecData.resolution')
            ev = sd.resolution @ ev
        polys = get_poly_basis(sd.lam, npoly, rbf=rbf)
        if espec_systematic is not None:
            es = np.sqrt(espec_systematic**2 + sd.espec**2)
        else:
            es = sd.espec
        f0 = get_chisq0_c if use_c else (
            lambda s, t, p, e, get_coeffs=False: get_chisq0(
                s, t, p, get_coeffs=get_coeffs, espec=e))
        cur = f0(sd.spec, ev, polys, es, get_coeffs=full_output)
        if full_output:
            cur, coeffs = cur
            model = np.dot(coeffs, polys * ev)
            out['raw_models'].append(ev)
            out['models'].append(model)
            dev = (model - sd.spec) / sd.espec
            good = ~sd.badmask
            tc = np.sum(dev[good]**2)
            out['chisq_array'].append(tc)
            out['npix_array'].append(good.sum())
            out['red_chisq_array'].append(tc / good.sum())
        if not np.isfinite(float(cur)):
            if outside > 0 and np.isfinite(ev).all():
                continue
            raise RuntimeError('non finite log-likelihood')
        accum += float(cur)
    if full_output:
        out['chisq'] = accum
        out['logl'] = -0.5 * accum
        return out
    return accum


# --------------------------------------------------------------------------
# A12  find_best + moments   (spec_fit.py:992-1092)
# --------------------------------------------------------------------------
def quadratic_interp_min(vel_grid, chisq, i):
    if i == 0 or i == len(vel_grid) - 1:
        return vel_grid[i]
    a2, a1, _ = np.polyfit(vel_grid[i - 1:i + 2], chisq[i - 1:i + 2], 2)
    val = -a1 / 2 / a2
    assert vel_grid[i - 1] < val < vel_grid[i + 1]
    return val


def grid_summary(vel_grid, chisq, quadratic=True):
    """chisq [Nv, Np] -> the find_best result dict (without best_param)."""
    i1, i2 = np.unravel_index(np.argmin(chisq), chisq.shape)
    probs = np.exp(-0.5 * (chisq[:, i2] - chisq[i1, i2]))
    probs = probs / probs.sum()
    bv = quadratic_interp_min(vel_grid, chisq[:, i2], i1) if quadratic else vel_grid[i1]
    err = np.sqrt((probs * (vel_grid - bv)**2).sum())
    if err < 1e-10:
        kur, skw = 0, 0
    else:
        kur = (probs * (vel_grid - bv)**4).sum() / err**4
        skw = (probs * (vel_grid - bv)**3).sum() / err**3
    return dict(best_chi=chisq[i1, i2], best_vel=bv, vel_err=err, i2=int(i2),
                kurtosis=kur, skewness=skw, probs=probs)


def chisq_grid(specdata, vel_grid, params_list, rot_params, options, config,
               libs, use_c=True, resol_params=None):
    cache = {}
    grid = np.zeros((len(vel_grid), len(params_list)))
    for j, p in enumerate(params_list):
        for i, v in enumerate(vel_grid):
            grid[i, j] = get_chisq(specdata, v, p, rot_params, options=options,
                                   config=config, libs=libs, cache=cache,
                                   use_c=use_c, resol_params=resol_params)
    return grid


def chisq_grid_fast(specdata, vel_grid, params, rot_params, options, config,
                    libs):
    """Same numbers as chisq_grid for ONE template, with the velocity loop in C
    (orc_chisq_vel).  Used as the CPU baseline and for large parity cases."""
    npoly = options.get('npoly') or 5
    rbf = options.get('rbf_continuum', True)
    vel_grid = _c(vel_grid)
    tot = np.zeros(len(vel_grid))
    badchi = 10 * sum(len(_.lam) for _ in specdata)
    for sd in specdata:
        lib = libs[sd.name]
        outside, tspec = get_cur_templ(lib, params, rot_params)
        if not np.isfinite(outside):
            tot += 1000 * badchi
            continue
        tot += outside * badchi
        _overlap_check(lib.lam[0], lib.lam[-1], sd.lam[0], sd.lam[-1],
                       min(config['min_vel'], vel_grid.min()),
                       max(config['max_vel'], vel_grid.max()))
        spl = Spline(lib.lam, tspec, log_step=lib.log_step)
        polys = _c(get_poly_basis(sd.lam, npoly, rbf=rbf))
        out = np.zeros(len(vel_grid))
        rc = core().orc_chisq_vel(_p(sd.lam), _p(sd.spec), _p(sd.espec),
                                  _p(polys), npoly, len(sd.lam), _p(spl.xs),
                                  _p(spl.h), _p(spl.A), _p(spl.B), _p(spl.C),
                                  _p(spl.D), spl.N, spl.log_step, _p(vel_grid),
                                  len(vel_grid), _p(out))
        assert rc == 0
        tot += out
    return tot


def find_best(specdata, vel_grid, params_list, rot_params=None, options=None,
              config=None, libs=None, quadratic=True, use_c=True,
              resol_params=None):
    grid = chisq_grid(specdata, vel_grid, params_list, rot_params, options,
                      config, libs, use_c=use_c, resol_params=resol_params)
    ret = grid_summary(vel_grid, grid, quadratic=quadratic)
    ret['best_param'] = params_list[ret.pop('i2')]
    ret['chisq_grid'] = grid
    return ret


# --------------------------------------------------------------------------
# A13  continuum-only chi^2   (spec_fit.py:739-783)
# --------------------------------------------------------------------------
def get_chisq_continuum(specdata, options=None):
    npoly = options.get('npoly') or 5
    rbf = options.get('rbf_continuum', True)
    ca, ra = np.zeros(len(specdata)), np.zeros(len(specdata))
    for i, sd in enumerate(specdata):
        polys = get_poly_basis(sd.lam, npoly, rbf=rbf)
        templ = np.ones(len(sd.spec))
        if sd.resolution is not None:  # spec_fit.py:765-767
            templ = sd.resolution @ templ
        _, co = get_chisq0(sd.spec, templ, polys, get_coeffs=True,
                           espec=sd.espec)
        dev = (np.dot(co, polys * templ) - sd.spec) / sd.espec
        good = ~sd.badmask
        ca[i] = np.sum(dev[good]**2)
        ra[i] = ca[i] / good.sum()
    return dict(chisq_array=ca, redchisq_array=ra)


# --------------------------------------------------------------------------
# A15  CCF pre-processing   (make_ccf.py:105-164, 288-414)
# --------------------------------------------------------------------------
def interp_masker(lam, spec, badmask):
    out = spec * 1
    bad = np.nonzero(badmask)[0]
    good = np.nonzero(~badmask)[0]
    if len(good) == 0:
        out[~np.isfinite(out)] = 1
        return out
    k = np.searchsorted(good, bad)
    le, re = k == 0, k == len(good)
    mid = ~le & ~re
    l1, l2 = lam[good[k[mid] - 1]], lam[good[k[mid]]]
    s1, s2 = spec[good[k[mid] - 1]], spec[good[k[mid]]]
    l0 = lam[bad[mid]]
    out[bad[le]] = spec[good[0]]
    out[bad[re]] = spec[good[-1]]
    out[bad[mid]] = (-(l1 - l0) * s2 + (l2 - l0) * s1) / (l2 - l1)
    return out


def continuum_nodes(lam0, splinestep):
    lammin = lam0.min()
    dl = np.log(1 + splinestep / 3e5)
    N = int(np.ceil(np.log(lam0.max() / lammin) / dl))
    nodes = lammin * np.exp(np.arange(N) * dl)
    edges = lammin * np.exp((-0.5 + np.arange(N + 1)) * dl)
    return nodes, edges


def continuum_model(p, nodes, lam):
    return np.exp(np.clip(
        scipy.interpolate.UnivariateSpline(nodes, p, s=0, k=2)(lam), -100, 100))


def continuum_start(lam0, spec0, edges):
    med = np.median(spec0)
    if med <= 0:
        med = np.abs(med)
        if med == 0:
            med = 1
    with np.errstate(all='ignore'):
        bs = scipy.stats.binned_statistic(lam0, spec0, 'median', bins=edges)
        p0 = np.log(np.maximum(bs.statistic, 1e-3 * med))
    p0[~np.isfinite(p0)] = np.log(med)
    return p0


def get_continuum(lam0, spec0, espec0, ccfconf, details=False):
    nodes, edges = continuum_nodes(lam0, ccfconf['splinestep'])
    p0 = continuum_start(lam0, spec0, edges)

    def resid(p):
        return (continuum_model(p, nodes, lam0) - spec0) / espec0

    sol = scipy.optimize.least_squares(resid, p0, loss='soft_l1')
    cont = continuum_model(sol['x'], nodes, lam0)
    if details:
        return cont, dict(p0=p0, x=sol['x'], cost=sol['cost'], nodes=nodes)
    return cont


def preprocess_data(lam, spec0, espec, ccfconf, badmask=None, maxerr=10,
                    details=False):
    ccf_logl = np.linspace(ccfconf['logl0'], ccfconf['logl1'],
                           ccfconf['npoints'])
    ccf_lam = np.exp(ccf_logl)
    ce, cs = espec.copy(), spec0.copy()
    if badmask is None:
        badmask = np.zeros(len(ce), dtype=bool)
    filt = scipy.signal.medfilt(cs, 11)
    mederr = np.nanmedian(ce)
    if ccfconf['continuum']:
        badmask = badmask | (ce > maxerr * mederr) | (filt <= 0)
    ce[badmask] = 1e9 * mederr
    cs = interp_masker(lam, cs, badmask)
    info = {}
    if ccfconf['continuum']:
        cont, info = get_continuum(lam, cs, ce, ccfconf, details=True)
    else:
        cont = 1
    ivar = 1. / ce**2
    ivar[badmask] = 0
    medv = np.median(cs)
    cont = np.maximum(1e-2 * medv, cont) if medv > 0 else np.maximum(cont, 1)
    c_spec = spec0 / cont
    ivar = cont**2 * ivar
    c_spec[badmask] = 0
    xind = np.searchsorted(lam, ccf_lam) - 1
    sub = (xind >= 0) & (xind <= (len(lam) - 2))
    r1, r2 = np.zeros(len(ccf_logl)), np.zeros(len(ccf_logl))
    li = xind[sub]
    ri = li + 1
    rw = (ccf_lam[sub] - lam[li]) / (lam[ri] - lam[li])
    lw = 1 - rw
    r1[sub] = lw * c_spec[li] + rw * c_spec[ri]
    a, b = ivar[li], ivar[ri]
    r2[sub] = a * b / (lw**2 * b + rw**2 * a + ((a * b) == 0).astype(int))
    if details:
        info.update(badmask=badmask, filled=cs, cespec=ce, cont=cont,
                    xind=xind)
        return r1, r2, info
    return r1, r2


# --------------------------------------------------------------------------
# A14  CCF fit   (fitter_ccf.py:62-253)
# --------------------------------------------------------------------------
def ccf_lag_tables(logl0, logl1, npoints, maxvel):
    step = (np.exp((logl1 - logl0) / npoints) - 1) * 3e5
    L = npoints
    off = L // 2
    vels = -((np.arange(L) + off) % L - off) * step
    sel = np.abs(vels) < (maxvel + step)
    assert sel.sum() % 2 == 1
    ind = np.roll(np.nonzero(sel)[0], sel.sum() // 2)[::-1]
    sub = vels[ind]
    if not np.all(np.diff(sub) > 0):
        raise RuntimeError('Velocity grid for CCF interpolation is invalid')
    return step, ind, sub


def _lin_interp_rows(x, Y, xnew):
    """scipy.interpolate.interp1d(kind='linear', assume_sorted=True) on rows."""
    hi = np.clip(np.searchsorted(x, xnew), 1, len(x) - 1)
    lo = hi - 1
    sl = (Y[:, hi] - Y[:, lo]) / (x[hi] - x[lo])[None, :]
    return sl * (xnew - x[lo])[None, :] + Y[:, lo]


def ccf_fit(specdata, config, libs, details=False):
    maxvel = config.get('max_vel') or 1000
    nvel = 2 * int(maxvel * 1. / (config.get('vel_step0') or 2)) + 1
    vel_grid = np.linspace(-maxvel, maxvel, nvel)
    total_sse = 0
    states, proc = [], {}
    for sd in specdata:
        cc = _ccf_pick(libs[sd.name], config)
        ps, pi = preprocess_data(sd.lam, sd.spec, sd.espec, cc,
                                 badmask=sd.badmask)
        proc[sd.name] = (ps, pi)
        total_sse += (ps**2 * pi).sum()
        step, ind, sub = ccf_lag_tables(cc['logl0'], cc['logl1'],
                                        cc['npoints'], maxvel)
        states.append(dict(S=np.fft.rfft(ps * pi).conj(),
                           V=np.fft.rfft(pi).conj(), ind=ind, sub=sub,
                           step=step, cc=cc))
    ref = states[0]['cc']
    for st in states[1:]:
        if (not np.array_equal(ref['params'], st['cc']['params'])
                or not np.array_equal(ref['vsinis'], st['cc']['vsinis'],
                                      equal_nan=True)):
            raise RuntimeError('The parameters of the CCF templates do not match')
    T = ref['fft'].shape[0]
    allchi = np.zeros((T, nvel))
    for st in states:
        cc = st['cc']
        c0 = np.fft.irfft(cc['fft'] * st['S'][None, :], axis=1)
        c1 = np.fft.irfft(cc['fft2'] * st['V'][None, :], axis=1)
        chi = -2 * c0 + c1 if cc['continuum'] else -c0**2 / c1
        allchi += _lin_interp_rows(st['sub'], chi[:, st['ind']], vel_grid)
    allchi += total_sse
    best_id = int(np.argmin(allchi.min(axis=1)))
    best_ccf = allchi[best_id]
    bp = int(np.argmin(best_ccf))
    if bp not in (0, len(best_ccf) - 1):
        co = np.polyfit(vel_grid[bp - 1:bp + 2], best_ccf[bp - 1:bp + 2], 2)
        best_vel = -co[1] / (2 * co[0]) if co[0] > 0 else vel_grid[bp]
    else:
        best_vel = vel_grid[bp]
    if not np.isfinite(allchi[best_id, bp]):
        raise RuntimeError('Cross-correlation step failed')
    best_model = {
        sd.name: np.roll(st['cc']['mod'][best_id], int(best_vel / st['step']))
        for sd, st in zip(specdata, states)
    }
    res = dict(best_par=ref['params'][best_id], best_vel=best_vel,
               best_ccf=best_ccf, best_vsini=ref['vsinis'][best_id],
               best_model=best_model, best_id=best_id, vel_grid=vel_grid,
               proc_spec={k: v[0] for k, v in proc.items()},
               proc_ivar={k: v[1] for k, v in proc.items()})
    if details:
        res['all_chisqs'] = allchi
    return res


# --------------------------------------------------------------------------
# A16  chi^2-grid callers   (vel_fit.py:13-94, 315-439)
# --------------------------------------------------------------------------
def minimum_sampler(func, best_vel, min_vel, max_vel, vel_step0, min_vel_step,
                    crit_ratio=5, goal_width=10):
    vel_step = vel_step0
    grids = []
    for it in range(10):
        vg = np.arange(
            math.ceil((min_vel - best_vel) / vel_step) * vel_step,
            max_vel - best_vel, vel_step) + best_vel
        grids.append(vg)
        best_vel, cur_err, res1 = func(vg)
        if vel_step < cur_err / crit_ratio or vel_step < min_vel_step:
            break
        if vel_step > cur_err:
            new_step, width = vel_step / crit_ratio, vel_step * goal_width
        else:
            new_step, width = cur_err / crit_ratio * 0.8, cur_err * goal_width
        min_vel = max(best_vel - width, min_vel)
        max_vel = min(best_vel + width, max_vel)
        vel_step = new_step
    return best_vel, cur_err, res1, grids


def find_best_vel_iterate(best_vel, config, specdata, params, rot_params,
                          options, libs):
    min_vel, max_vel = config['min_vel'], config['max_vel']
    best_vel = min(max(best_vel, min_vel), max_vel)

    def func(vg):
        tot = chisq_grid_fast(specdata, vg, params, rot_params, options,
                              config, libs)
        r = grid_summary(vg, tot[:, None])
        return r['best_vel'], r['vel_err'], r

    bv, be, r, grids = minimum_sampler(func, best_vel, min_vel, max_vel,
                                       config['vel_step0'],
                                       config['min_vel_step'])
    return bv, be, r['skewness'], r['kurtosis'], grids


def firstguess(specdata, options, config, libs, vsinigrid=(None, 10, 100),
               paramsgrid=None):
    if paramsgrid is None:
        paramsgrid = {'logg': [1, 2, 3, 4, 5], 'teff': [3000, 5000, 8000, 10000],
                      'feh': [-2, -1, 0], 'alpha': [0]}
    names = libs[specdata[0].name].parnames
    params = []
    for x in itertools.product(*paramsgrid.values()):
        d = dict(zip(paramsgrid.keys(), x))
        params.append([d[_] for _ in names])
    vg = np.arange(config['min_vel'], config['max_vel'], config['vel_step0'])
    best, bestpar = np.inf, None
    for vs in vsinigrid:
        rot = None if vs is None else (vs, )
        cols = [chisq_grid_fast(specdata, vg, p, rot, options, config, libs)
                for p in params]
        r = grid_summary(vg, np.array(cols).T)
        if r['best_chi'] < best:
            bestpar = dict(zip(names, params[r['i2']]))
            if vs is not None:
                bestpar['vsini'] = vs
            best = r['best_chi']
    return bestpar


# --------------------------------------------------------------------------
# A4  NN template evaluator   (nn/NNInterpolator.py:14-91, 159-171;
#                              nn/RVSInterpolator.py:36-42)
# --------------------------------------------------------------------------
def silu32(x):
    x = x.astype(np.float32)
    return (x / (np.float32(1) + np.exp(-x))).astype(np.float32)


def nn_forward(weights, p, M, S, log_ids=(0, )):
    """weights: list of (W[out,in] f32, b[out] f32); SiLU after every layer but
    the last.  Returns exp(clip(float64(out), -300, 300))."""
    x1 = np.asarray(p, dtype=np.float32)
    y = x1 * 1
    with np.errstate(all='ignore'):
        for i in log_ids:
            y[..., i] = np.log10(x1[..., i])
    h = ((y - M) / S).astype(np.float32).reshape(-1, len(M))
    for li, (W, b) in enumerate(weights):
        h = (h @ W.T.astype(np.float32) + b.astype(np.float32)).astype(np.float32)
        if li < len(weights) - 1:
            h = silu32(h)
    return np.exp(np.clip(h.astype(np.float64), -300, 300))


class NNLibrary(Library):
    """A spectral setup whose template evaluator is the MLP (interpolation_type
    'generic' with nn.RVSInterpolator / nn.OutsideInterpolator,
    spec_inter.py:371-378, nn/RVSInterpolator.py:36-71); the CCF part of the
    setup is that of Library."""

    def __init__(self, npz):
        d = dict(npz)
        self.lam = _c(d['lam'])
        self.log_step = bool(d['log_step'])
        self.log_ids = [int(_) for _ in np.atleast_1d(d['log_ids'])]
        self.parnames = tuple(str(_) for _ in d['parnames'])
        dims = [int(_) for _ in d['nn_dims']]
        self.ndim = dims[0]
        self.weights = [(np.asarray(d['nn_W%d' % i], dtype=np.float32),
                         np.asarray(d['nn_b%d' % i], dtype=np.float32))
                        for i in range(len(dims) - 1)]
        self.M, self.S = _c(d['nn_M']), _c(d['nn_S'])
        self.hull = None
        if 'nn_pts' in d:
            import scipy.spatial
            pts = np.asarray(d['nn_pts'], dtype=np.float64)
            self.hull = (scipy.spatial.ConvexHull(pts[:, :2]).equations,
                         scipy.spatial.ConvexHull(pts[:, 2:]).equations)
        self.ccf_sets = _ccf_sets(d)
        self.ccf = self.ccf_sets.get(True)

    def eval(self, p, details=False):
        spec = np.ascontiguousarray(
            nn_forward(self.weights, np.asarray(p, dtype=np.float64)[None],
                       self.M, self.S, self.log_ids)[0], dtype=np.float64)
        return (spec, dict()) if details else spec

    # nn/RVSInterpolator.py:63-71 on the Mapper-transformed point
    def outside_flag(self, p):
        if self.hull is None:
            return 0.
        x1 = np.asarray(p, dtype=np.float32)
        y = x1 * 1
        with np.errstate(all='ignore'):
            for i in self.log_ids:
                y[i] = np.log10(x1[i])
        q = (y.astype(np.float64) - self.M) / self.S
        dx = (self.hull[0][:, :-1] @ q[:2] + self.hull[0][:, -1]).max()
        dy = (self.hull[1][:, :-1] @ q[2:] + self.hull[1][:, -1]).max()
        return max(dx, dy, 0.)**2


def make_library(npz):
    """Library / NNLibrary / TriLibrary by the keys of the converted artefact"""
    if 'nn_dims' in npz:
        return NNLibrary(npz)
    if 'simplices' in npz:
        return TriLibrary(npz)
    return Library(npz)


def convolve_vsini_rows(lam, templ, vsini, eps=0.6):
    """convolve_vsini applied to every row (bench helper)."""
    return np.array([convolve_vsini(lam, t, v, eps) for t, v in zip(templ, vsini)])


# --------------------------------------------------------------------------
# vel_fit.process (vel_fit.py:95-312, 442-737)
# --------------------------------------------------------------------------
def _vsini_to_vsini(x, max_vsini):  # VSiniMapper.to_vsini, vel_fit.py:108-116
    vsini = np.clip(x, 0, max_vsini)
    pen = int(x < 0) * (vsini - x)**2 + int(x > max_vsini) * (vsini - x)**2
    return vsini, pen


def _pm_forward(p0, names, pd0, fix, fit_vsini, max_vsini):
    """ParamMapper.forward, vel_fit.py:156-194"""
    rev = list(p0)[::-1]
    ret = dict(vel=rev.pop())
    pen = 0
    if fit_vsini:
        ret['vsini'], pen = _vsini_to_vsini(rev.pop(), max_vsini)
    else:
        ret['vsini'] = pd0['vsini'] if 'vsini' in fix else None
    ret['rot_params'] = None if ret['vsini'] is None else (ret['vsini'], )
    ret['params'] = [pd0[x] if x in fix else rev.pop() for x in names]
    assert len(rev) == 0
    ret['penalty'] = pen
    return ret


def hessian_central(f, x, h):
    """the stand-in for numdifftools.Hessian used by the golden harness
    (tests/golden/make_golden_process.py): central second differences, one step"""
    x = np.asarray(x, dtype=float)
    n = len(x)
    fx = f(x)
    ee = np.diag(h)
    H = np.zeros((n, n))
    for i in range(n):
        H[i, i] = (f(x + 2 * ee[i]) - 2 * fx + f(x - 2 * ee[i])) / \
            (4. * h[i] * h[i])
        for j in range(i + 1, n):
            H[i, j] = (f(x + ee[i] + ee[j]) - f(x + ee[i] - ee[j]) -
                       f(x - ee[i] + ee[j]) + f(x - ee[i] - ee[j])) / \
                (4. * h[i] * h[j])
            H[j, i] = H[i, j]
    return H


def uncertainties_from_hessian(hessian):  # vel_fit.py:464-502
    import scipy.linalg
    dh = np.diag(hessian)
    with np.errstate(all='ignore'):
        inv_d = 1. / (dh + (dh == 0))
    inv_d[dh == 0] = np.inf
    bad = False
    try:
        hinv = scipy.linalg.inv(hessian)
    except (np.linalg.LinAlgError, ValueError):
        bad = True
        hinv = np.diag(inv_d)
    e0 = np.array(np.diag(hinv))
    e1 = inv_d
    b0, b1 = e0 < 0, e1 < 0
    if b0.any():
        bad = True
    s1, s2 = b0 & ~b1, b0 & b1
    e0[s1] = e1[s1]
    e0[s2] = 0
    with np.errstate(all='ignore'):
        err = np.sqrt(e0)
    err[s2] = np.nan
    if (~np.isfinite(err)).sum() != 0:
        bad = True
    return err, hinv, bad


def process(specdata, paramDict0, fixParam, options, config, libs,
            priors=None):
    """vel_fit.process (vel_fit.py:505-737) without the optional BFGS polish;
    Nelder-Mead is scipy's own (as in the reference); the Hessian is the
    stand-in rule above (numdifftools is absent)."""
    import scipy.optimize
    names = list(libs[specdata[0].name].parnames)
    fix = list(fixParam or [])
    min_vel, max_vel = config['min_vel'], config['max_vel']
    max_vsini = config['max_vsini']
    curparam = tuple(paramDict0[_] for _ in names)
    if 'vsini' not in paramDict0:
        rot, fit_vsini = None, False
    else:
        rot = (paramDict0['vsini'], )
        fit_vsini = 'vsini' not in fix
    vg = np.arange(min_vel, max_vel, config['vel_step0'])
    res = find_best(specdata, vg, [curparam], rot_params=rot, options=options,
                    config=config, libs=libs)
    # _get_simplex_start, vel_fit.py:272-312
    start, std = [res['best_vel']], [5]
    if fit_vsini:
        start.append(np.clip(paramDict0['vsini'], 0, max_vsini))
        std.append(3)
    for x in names:
        if x not in fix:
            start.append(paramDict0[x])
            std.append({'logg': 0.5, 'teff': 300, 'feh': 0.5,
                        'alpha': 0.25}.get(x) or 0.5)
    curval = np.array(start, dtype=float)
    nd = len(curval)
    R = np.random.RandomState(43434)
    simplex = np.zeros((nd + 1, nd))
    simplex[0] = curval
    simplex[1:] = curval[None, :] + np.array(std)[None, :] * R.normal(
        size=(nd, nd))
    cache = {}

    def chisq0(pd):  # chisq_func0, vel_fit.py:205-226
        c = 0
        if priors is not None:
            for i, k in enumerate(names):
                if k in priors:
                    c += ((priors[k][0] - pd['params'][i]) / priors[k][1])**2
        cache.clear()
        return c + get_chisq(specdata, pd['vel'], tuple(pd['params']),
                             pd['rot_params'], options=options, config=config,
                             libs=libs, use_c=True)

    def chisq_func(p):  # vel_fit.py:229-254
        pd = _pm_forward(p, names, paramDict0, fix, fit_vsini, max_vsini)
        if (pd['vel'] > max_vel or pd['vel'] < min_vel
                or (~np.isfinite(pd['params'])).any()):
            return 1e30
        return chisq0(pd) + pd['penalty']

    nits, nfevs = [], []
    success = True
    for it in range(2):
        r0 = scipy.optimize.minimize(
            chisq_func, curval, method='Nelder-Mead',
            options=dict(fatol=1e-3, xatol=1e-2, initial_simplex=simplex,
                         maxiter=10000, maxfev=np.inf))
        curval, simplex = r0['x'], r0['final_simplex'][0]
        nits.append(r0['nit'])
        nfevs.append(r0['nfev'])
        if r0['success']:
            break
        if it == 1:
            success = False
    best = _pm_forward(r0['x'], names, paramDict0, fix, fit_vsini, max_vsini)
    bv, be, sk, ku, _ = find_best_vel_iterate(
        best['vel'], config, specdata, tuple(best['params']),
        best['rot_params'], options, libs)
    outp = get_chisq(specdata, bv, tuple(best['params']), best['rot_params'],
                     options=options, config=config, libs=libs,
                     full_output=True, use_c=True)
    base = np.array([{'vsini': 1 / 100, 'logg': 0.1 / 100, 'feh': 0.1 / 100,
                      'alpha': .01 / 100, 'teff': 1 / 100,
                      'vrad': 1 / 100}[_] for _ in names])
    tmp = dict(best)

    def hess_func(p):  # vel_fit.py:257-269
        tmp['params'] = list(p)
        return 0.5 * chisq0(tmp)

    # vel_fit.py:710-725 with oracle/numdiff_restated.py standing for the
    # numdifftools package: MinStepGenerator(base_step) first, the default
    # generator (step=None) if the result is flagged
    from . import numdiff_restated as ndf
    x = np.array(best['params'], dtype=float)
    step_gen = ndf.MinStepGenerator(base_step=base)
    for _ in range(2):
        H = ndf.Hessian(hess_func, step=step_gen)(x)
        err, covar, bad = uncertainties_from_hessian(H)
        if bad:
            step_gen = None
    return dict(param=dict(zip(names, best['params'])), vsini=best['vsini'],
                vel=bv, vel_err=be, vel_skewness=sk, vel_kurtosis=ku,
                nm_x=r0['x'], nm_fun=r0['fun'], nm_nit=nits, nm_nfev=nfevs,
                param_err=dict(zip(names, err)), param_covar=covar,
                minimize_success=success, bad_hessian=bad,
                chisq=outp['chisq'], chisq_array=outp['chisq_array'],
                npix_array=outp['npix_array'], yfit=outp['models'])
