"""Batched device engine: packs spectra in HBM and drives the HIP kernels.

Host code is Python; every numerical step is a call into librvsgpu.so
(include/rvsgpu.h).  PyTorch is used only for device memory, streams and
small index glue (gather of CCF parameters, reshapes).

HBM layout (per arm of a batch of S spectra, all sharing the wavelength grid):
    lam      float64 [npix]
    spec     float64 [S, npix]      espec float64 [S, npix]   badmask uint8 [S, npix]
    polysT   float64 [npix, npoly]  continuum basis, pixel-major (scalar-cache reads)
    work     float64 [npix + 2*S*npix + 2*S]  velocity independent per-pixel terms
    templ    float64 [J, ntp]   ->  coef float64 [J, ntp, 4]  spline records
    chisq    float64 [J, Nv]
"""
import numpy as np
import torch

from . import _lib
from . import ccf_tables

SPEED_OF_LIGHT = 299792.458  # km/s, spec_fit.py:23

# optional per-kernel timing with HIP events on the launch stream (bench.py):
# KTIMERS = {} enables it; each entry is a list of (start, end, units) tuples.
KTIMERS = None


class _ktime:

    def __init__(self, name, units=1):
        self.name, self.units = name, units

    def __enter__(self):
        if KTIMERS is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if KTIMERS is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            KTIMERS.setdefault(self.name, []).append((self.e0, e1, self.units))


def ktimers_summary():
    """name -> (n_launch_groups, total_ms, total_units) after a synchronize"""
    out = {}
    for k, lst in (KTIMERS or {}).items():
        ms = sum(a.elapsed_time(b) for a, b, _ in lst)
        out[k] = (len(lst), ms, sum(u for _, _, u in lst))
    return out


def get_poly_basis(lam, npoly, rbf=True):
    """Continuum basis, spec_fit.py:148-176 (host, float64; depends only on the
    arm's wavelength grid so it is built once per arm and uploaded)."""
    lam = np.asarray(lam, dtype=np.float64)
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


# per-grid tables (basis, orthonormal basis, CCF rebin / spline tables) on the device
# (csrc/tables.hip); False: numpy, grid by grid (tests/test_tables_gpu.py compares)
DEVICE_TABLES = True

RES_MAXND = 33   # widest band the velocity-grid kernel holds in its LDS ring;
                 # wider matrices run the grid through the point kernel


def resol_taps(mats, npix):
    """A9: banded resolution matrices (scipy.sparse, e.g. the dia matrices of
    construct_resol_mat / desi_fit.construct_resolution_sparse_matrix) as row
    taps for the kernels: taps[s, k, d] = R_s[k, k - m + d], m = (nd-1)/2.
    A matrix may be SMALLER than npix (a spectrum on a shorter grid of a grid
    set): its rows fill the first pixels, everything behind them is zero.
    Returns (taps float64 [len(mats), npix, nd], nd)."""
    import scipy.sparse
    m = 0
    dias = []
    for M in mats:
        D = scipy.sparse.dia_matrix(M)
        assert D.shape[0] == D.shape[1] and D.shape[0] <= npix
        offs = [int(o) for o, row in zip(D.offsets, D.data) if np.any(row != 0)]
        if offs:
            m = max(m, max(abs(o) for o in offs))
        dias.append(D)
    nd = 2 * m + 1
    taps = np.zeros((len(mats), npix, nd))
    for i, D in enumerate(dias):
        n = D.shape[0]
        k = np.arange(n)
        for o, row in zip(D.offsets, D.data):
            o = int(o)
            if abs(o) > m:
                continue
            # dia storage: data[d, j] = R[j - o, j]  ->  R[k, k + o] = row[k + o]
            ok = (k + o >= 0) & (k + o < n)
            taps[i, k[ok], m + o] += row[k[ok] + o]
    return taps, nd


class ArmData:
    """One spectral arm of a batch: S spectra on a common wavelength grid -- or,
    with `grid_id`, on G grids of their own (the reference takes any `lam` per
    object, spec_fit.py:70-145; SDSS-style spectra, tests/test_sdss.py): `lam` is
    then a list of G wavelength arrays and grid_id[s] the grid of spectrum s; spec
    / espec / badmask are [S, npix] with npix the longest grid, a spectrum on a
    shorter grid padded behind its last pixel (any spec, espec and badmask are
    overwritten there).  The kernels read the grid, the pixel coordinates and the
    continuum basis of a job's own grid (include/rvsgpu.h, "grid sets").
    resolution: None, or a list of S scipy.sparse matrices (SpecData.resolution
    .mat, spec_fit.py:54-67) applied to the resampled template (A9)."""

    def __init__(self, name, lam, spec, espec, badmask=None, device='cuda',
                 resolution=None, grid_id=None):
        _lib.require_gpu()
        self.name = name
        self.device = device
        if grid_id is None:
            self.G = 1
            self.lam_host = np.ascontiguousarray(lam, dtype=np.float64)
            self.npix = len(self.lam_host)
            self.npix_g = np.array([self.npix], dtype=np.int32)
            self.grids = [self.lam_host]
            self.grid_id = self.grid_id_host = None
            self.lam = torch.as_tensor(self.lam_host).to(device)
        else:
            self.grids = [np.ascontiguousarray(g, dtype=np.float64) for g in lam]
            self.G = len(self.grids)
            self.npix_g = np.array([len(g) for g in self.grids], dtype=np.int32)
            self.npix = int(self.npix_g.max())
            lam2 = np.empty((self.G, self.npix))
            for i, g in enumerate(self.grids):   # padding repeats the last pixel
                lam2[i, :len(g)] = g
                lam2[i, len(g):] = g[-1]
            # (lam_host: the longest grid, for code that only needs the arm's
            # extent -- overlap checks use grid_extent())
            self.lam_host = self.grids[int(np.argmax(self.npix_g))]
            self.grid_id_host = np.ascontiguousarray(grid_id, dtype=np.int32)
            assert self.grid_id_host.min() >= 0 and \
                self.grid_id_host.max() < self.G
            self.grid_id = torch.as_tensor(self.grid_id_host).to(device)
            self.lam = torch.as_tensor(lam2).to(device)
        self.spec = self._as2d(spec, torch.float64)
        self.espec = self._as2d(espec, torch.float64)
        self.S = self.spec.shape[0]
        if badmask is None:
            self.badmask = torch.zeros((self.S, self.npix), dtype=torch.uint8,
                                       device=device)
        else:
            self.badmask = self._as2d(badmask, torch.uint8)
        assert self.spec.shape == self.espec.shape == self.badmask.shape
        assert self.spec.shape[1] == self.npix
        if self.G > 1:
            assert len(self.grid_id_host) == self.S
            # the padding: no weight (espec = +inf is the kernels' marker), masked
            n_s = torch.as_tensor(self.npix_g).to(device)[self.grid_id.long()]
            pad = torch.arange(self.npix, device=device)[None, :] >= n_s[:, None]
            if bool(pad.any()):
                self.spec = torch.where(pad, torch.zeros_like(self.spec), self.spec)
                self.espec = torch.where(pad, torch.full_like(self.espec, np.inf),
                                         self.espec)
                self.badmask = torch.where(pad, torch.ones_like(self.badmask),
                                           self.badmask)
        self.pen_scale = None   # (set by the SpecBatch that holds the arm)
        self._basis = {}
        self._work = {}
        self._ccf = {}
        self.resol = None
        if resolution is not None:
            self.set_resolution(resolution)

    def set_resolution(self, mats):
        assert len(mats) in (1, self.S)
        if self.G > 1:   # (one matrix per spectrum, of its own grid's size)
            assert len(mats) == self.S and all(
                m.shape[0] == self.npix_g[g]
                for m, g in zip(mats, self.grid_id_host))
        taps, nd = resol_taps(mats, self.npix)
        self.resol = make_resol(taps, nd, self.S, self.device)

    def subset(self, idx):
        """ArmData of the spectra idx (device long tensor)"""
        if self.G > 1:
            # only the grids the subset uses (a half of a 10 000-grid batch does not
            # carry the other half's tables)
            used, gid = np.unique(self.grid_id[idx].cpu().numpy(),
                                  return_inverse=True)
            npx = int(self.npix_g[used].max())
            if len(used) == 1:
                a = ArmData(self.name, self.grids[int(used[0])],
                            self.spec[idx][:, :npx], self.espec[idx][:, :npx],
                            self.badmask[idx][:, :npx], device=self.device)
            else:
                a = ArmData(self.name, [self.grids[int(k)] for k in used],
                            self.spec[idx][:, :npx], self.espec[idx][:, :npx],
                            self.badmask[idx][:, :npx], device=self.device,
                            grid_id=gid.astype(np.int32))
            if self.resol is not None:   # (per spectrum: taps [S, npix, nd])
                r = self.resol
                a.resol = dict(taps=r['taps'][idx][:, :npx].contiguous(),
                               nd=r['nd'], stride=npx * r['nd'],
                               unit=r['unit'][idx][:, :npx].contiguous())
            return a
        a = ArmData(self.name, self.lam_host, self.spec[idx], self.espec[idx],
                    self.badmask[idx], device=self.device)
        if self.resol is not None:
            r = self.resol
            if r['stride'] == 0:
                a.resol = dict(r)
            else:
                a.resol = dict(taps=r['taps'][idx].contiguous(), nd=r['nd'],
                               stride=r['stride'],
                               unit=r['unit'][idx].contiguous())
        return a

    def _as2d(self, a, dtype):
        if isinstance(a, torch.Tensor):
            t = a.to(device=self.device, dtype=dtype)
        else:
            # (uploaded in its own type, converted on the device: float32 survey
            # arrays cross the bus at half the bytes, the values are the same)
            t = torch.as_tensor(np.ascontiguousarray(a)).to(device=self.device)
            if t.dtype != dtype:
                t = t.to(dtype)
        if t.dim() == 1:
            t = t[None, :]
        return t.contiguous()

    def grid_extent(self):
        """(largest first wavelength, smallest last wavelength) over the grids"""
        return max(g[0] for g in self.grids), min(g[-1] for g in self.grids)

    def grid_args(self):
        """(grid_id pointer, G) of the _g entry points"""
        return (_lib.ptr(self.grid_id) if self.G > 1 else None), self.G

    def basis_stride(self, npoly):
        """doubles between the basis tables of consecutive grids"""
        return (self.npix + 1) * npoly

    def _npix_g_dev(self):
        if self.G == 1:
            return None
        if getattr(self, '_npix_g_t', None) is None:
            self._npix_g_t = torch.as_tensor(self.npix_g).to(self.device)
        return self._npix_g_t

    def _build_basis(self, npoly, rbf, ortho=True):
        """raw (and, with `ortho`, orthonormal) basis of every grid
        (rvs_basis_build), cached.  Returns (raw, Q^T, log-volume) or (raw,)."""
        key = ('dev', npoly, bool(rbf))
        if key in self._basis:
            return self._basis[key]
        rkey = ('devraw', npoly, bool(rbf))
        if not ortho and rkey in self._basis:
            return self._basis[rkey]
        G, npx = self.G, self.npix
        f64 = dict(dtype=torch.float64, device=self.device)
        raw = torch.empty((G, npx + 1, npoly), **f64)
        qt = torch.empty((G, npx + 1, npoly), **f64) if ortho else None
        ld = torch.empty(G, **f64) if ortho else None
        cen = torch.as_tensor(np.linspace(-1, 1, max(npoly - 3, 1), True)).to(
            self.device)
        rc = _lib.lib().rvs_basis_build(
            _lib.ptr(self.lam), _lib.ptr(self._npix_g_dev()), G, npx, npoly,
            int(bool(rbf)), _lib.ptr(cen), _lib.ptr(raw), _lib.ptr(qt),
            _lib.ptr(ld), _lib.stream())
        _lib.check(rc, 'rvs_basis_build')
        if ortho:
            self._basis[key] = (raw, qt, ld)
            self._basis.pop(rkey, None)
            return self._basis[key]
        self._basis[rkey] = (raw, )
        return self._basis[rkey]

    def basis(self, npoly, rbf):
        """pixel-major continuum basis get_poly_basis(lam).T (+ one zero row) of
        every grid: [npix + 1, npoly], or [G, npix + 1, npoly] (rows behind a
        grid's last pixel are zero)"""
        if DEVICE_TABLES:
            raw = self._build_basis(npoly, rbf, ortho=False)[0]
            return raw[0] if self.G == 1 else raw
        key = (npoly, bool(rbf))
        if key not in self._basis:
            PT = np.zeros((self.G, self.npix + 1, npoly))
            for i, g in enumerate(self.grids):
                PT[i, :len(g)] = get_poly_basis(g, npoly, rbf).T
            self._basis[key] = torch.as_tensor(
                PT[0] if self.G == 1 else PT).to(self.device)
        return self._basis[key]

    def basis_ortho(self, npoly, rbf):
        """The same function space in an orthonormal basis, for the chi^2-grid
        kernel: P^T = Q R over the pixels -> rows of Q^T.  The marginalised
        likelihood only depends on span(P): with M = R^T M' R,
            log det M = log det M' + 2 log|det R|     and  v^T M^-1 v = v'^T M'^-1 v',
        so the kernel works on M' (condition number ~1-10 instead of 1e4-1e5 for
        the monomial + RBF basis, which keeps the normal-equation rounding error
        of the in-register Cholesky at the 1e-10 level even at S/N 1000) and the
        constant 2 log|det R| is added back.  Returns (Q^T pixel-major, const):
        const is a float for one grid, a device tensor [S] (the constant of every
        spectrum's grid) for a grid set.  Built on the device (rvs_basis_build:
        modified Gram-Schmidt, twice) -- on the host a QR per grid is 0.4-8 ms, seconds
        to a minute for a batch of SDSS-style spectra; DEVICE_TABLES = False keeps
        numpy's Householder QR (tests compare the two)."""
        key = ('ortho', npoly, bool(rbf), DEVICE_TABLES)
        if key in self._basis:
            return self._basis[key]
        if DEVICE_TABLES:
            _, qt, ld = self._build_basis(npoly, rbf)
            if self.G == 1:
                self._basis[key] = (qt[0], float(ld[0].item()))
            else:
                self._basis[key] = (qt, ld[self.grid_id.long()])
            return self._basis[key]
        QT = np.zeros((self.G, self.npix + 1, npoly))
        off = np.zeros(self.G)
        for i, g in enumerate(self.grids):
            P = get_poly_basis(g, npoly, rbf)
            Q, R = np.linalg.qr(P.T)
            QT[i, :len(g)] = Q
            off[i] = 2.0 * float(np.sum(np.log(np.abs(np.diag(R)))))
        if self.G == 1:
            self._basis[key] = (torch.as_tensor(QT[0]).to(self.device),
                                float(off[0]))
        else:
            offs = torch.as_tensor(off).to(self.device)[self.grid_id.long()]
            self._basis[key] = (torch.as_tensor(QT).to(self.device), offs)
        return self._basis[key]

    def work(self, lib, espec_sys=0.0):
        """rvs_chisq_prepare output for this arm against the knots of `lib`."""
        key = (lib.name, id(lib), float(espec_sys))
        if key not in self._work:
            L = _lib.lib()
            n = L.rvs_chisq_work_size_g(self.npix, self.S, self.G)
            w = torch.empty(n, dtype=torch.float64, device=self.device)
            rc = L.rvs_chisq_prepare_g(_lib.ptr(self.lam), _lib.ptr(self.spec),
                                       _lib.ptr(self.espec), self.npix, self.S,
                                       self.G, _lib.ptr(lib.knots3),
                                       int(lib.log_step), float(espec_sys),
                                       _lib.ptr(w), _lib.stream())
            if rc == -3:
                raise AssertionError('spline knots are not uniformly spaced')
            _lib.check(rc, 'rvs_chisq_prepare')
            self._work[key] = w
        return self._work[key]

    def ccf_tables(self, lib, config):
        """Per-arm CCF tables (host-built once, see ccf_tables.py)."""
        cc = lib.ccf_set(config)
        maxvel, vgrid = ccf_tables.ccf_vel_grid(config)
        key = (lib.name, id(lib), maxvel, len(vgrid), cc['continuum'])
        if key in self._ccf:
            return self._ccf[key]
        dev = self.device
        nfft = cc['npoints']
        T = {}
        T['maxvel'], T['vgrid_host'] = maxvel, vgrid
        T['vgrid'] = torch.as_tensor(vgrid).to(dev)
        step, ind, sub = ccf_tables.lag_tables(cc['logl0'], cc['logl1'], nfft,
                                               maxvel)
        T['step'] = step
        L = _lib.lib()
        pos = np.array([L.rvs_ccf_fft_pos(nfft, int(n) >> 1) for n in ind],
                       dtype=np.int64)
        T['lag_pos'] = torch.as_tensor(
            (2 * pos + (ind & 1)).astype(np.int32)).to(dev)
        # output masks of the last two radix-8 passes (rvs_ccf_xcorr `prune`):
        # only these lags are read back from the inverse transform
        n2 = nfft // 2
        T['prune'] = None
        l2 = n2.bit_length() - 1
        if (1 << l2) == n2 and l2 % 3 == 0 and l2 >= 6 and XCORR_PRUNE:
            pm = np.zeros(n2 // 64 + n2 // 8, dtype=np.uint8)
            for p_ in pos:
                pm[n2 // 64 + (int(p_) >> 3)] |= 1 << (int(p_) & 7)
                pm[int(p_) >> 6] |= 1 << ((int(p_) >> 3) & 7)
            T['prune'] = torch.as_tensor(pm).to(dev)
        T['lag_vel'] = torch.as_tensor(sub).to(dev)
        T['nlag'] = len(ind)
        T['ilo'] = torch.as_tensor(ccf_tables.interp_tables(sub, vgrid)).to(dev)
        tw = np.exp(2j * np.pi * np.arange(nfft // 2) / nfft)
        T['twid'] = torch.as_tensor(
            np.ascontiguousarray(tw).view(np.float64)).to(dev)
        if DEVICE_TABLES:
            self._ccf_grid_tables_device(T, cc, nfft)
        else:
            self._ccf_grid_tables_host(T, cc, nfft)
        self._ccf[key] = T
        return T


    def _ccf_nodes(self, cc):
        """continuum nodes / bin edges of every grid (make_ccf.py:123-131), padded
        to the largest node count: nodes [G, nn], edges [G, nn + 1], count [G]"""
        l0 = np.array([g.min() for g in self.grids])
        l1 = np.array([g.max() for g in self.grids])
        dl = np.log(1 + cc['splinestep'] / 3e5)
        N = np.ceil(np.log(l1 / l0) / dl).astype(np.int64)
        nn = int(N.max())
        nodes = l0[:, None] * np.exp(np.arange(nn) * dl)[None, :]
        edges = l0[:, None] * np.exp((-0.5 + np.arange(nn + 1)) * dl)[None, :]
        return nodes, edges, N.astype(np.int32), nn

    def _ccf_grid_tables_device(self, T, cc, nfft):
        """per-grid tables by rvs_ccf_tables_build; the collocation matrices (a few
        dozen numbers per grid) on the host"""
        dev, G, npx = self.device, self.G, self.npix
        ccf_lam = torch.as_tensor(np.exp(np.linspace(cc['logl0'], cc['logl1'],
                                                     nfft))).to(dev)
        xi = torch.empty((G, nfft), dtype=torch.int32, device=dev)
        rw = torch.empty((G, nfft), dtype=torch.float64, device=dev)
        cont = bool(cc['continuum'])
        nodes_t = edges_t = nng_t = Eb = El = ist = bst = None
        nn = 0
        if cont:
            nodes, edges, nng, nn = self._ccf_nodes(cc)
            nodes_t = torch.as_tensor(nodes).to(dev)
            edges_t = torch.as_tensor(edges).to(dev)
            nng_t = torch.as_tensor(nng).to(dev)
            Eb = torch.empty((G, npx, 3), dtype=torch.float64, device=dev)
            El = torch.empty((G, npx), dtype=torch.int32, device=dev)
            ist = torch.empty((G, nn), dtype=torch.int32, device=dev)
            bst = torch.empty((G, nn + 1), dtype=torch.int32, device=dev)
        rc = _lib.lib().rvs_ccf_tables_build(
            _lib.ptr(self.lam), _lib.ptr(self._npix_g_dev()), G, npx,
            _lib.ptr(ccf_lam), nfft, int(cont), _lib.ptr(nodes_t),
            _lib.ptr(edges_t), _lib.ptr(nng_t), nn, _lib.ptr(xi), _lib.ptr(rw),
            _lib.ptr(Eb), _lib.ptr(El), _lib.ptr(ist), _lib.ptr(bst), _lib.stream())
        _lib.check(rc, 'rvs_ccf_tables_build')
        one = (G == 1)
        T['xind'], T['rw'] = (xi[0], rw[0]) if one else (xi, rw)
        T['npix_g'] = T['nnode_g'] = None
        if cont:
            Cinv = ccf_tables.collocation_batch(nodes, nng)
            T['Eb'], T['El'] = (Eb[0], El[0]) if one else (Eb, El)
            T['istart'], T['bin_start'] = (ist[0], bst[0]) if one else (ist, bst)
            T['Cinv'] = torch.as_tensor(Cinv[0] if one else Cinv).to(dev)
            T['nnode'] = nn
            if not one:
                T['nnode_g'] = nng_t
        else:
            T['Eb'] = T['El'] = T['Cinv'] = T['istart'] = None
            T['nnode'], T['bin_start'] = 0, None
        if G > 1:
            T['npix_g'] = self._npix_g_dev()

    def _ccf_grid_tables_host(self, T, cc, nfft):
        """the same tables with numpy, grid by grid (ccf_tables.py; DEVICE_TABLES =
        False: what rounds 1-3 shipped, kept as the statement the device tables are
        tested against)"""
        dev = self.device
        G = self.G
        xi = np.empty((G, nfft), dtype=np.int32)
        rw = np.empty((G, nfft))
        per = []
        for i, g in enumerate(self.grids):
            xi[i], rw[i] = ccf_tables.rebin_tables(g, cc['logl0'], cc['logl1'], nfft)
            if cc['continuum']:
                nodes, edges = ccf_tables.continuum_nodes(g, cc['splinestep'])
                per.append((nodes, edges) + ccf_tables.interp_spline_tables(nodes, g))
        T['xind'] = torch.as_tensor(xi if G > 1 else xi[0]).to(dev)
        T['rw'] = torch.as_tensor(rw if G > 1 else rw[0]).to(dev)
        T['npix_g'] = T['nnode_g'] = None
        if cc['continuum']:
            nn = max(len(q[0]) for q in per)
            npx = self.npix
            Eb = np.zeros((G, npx, 3))
            El = np.zeros((G, npx), dtype=np.int32)
            Cinv = np.zeros((G, 2 * nn * nn))
            istart = np.zeros((G, nn), dtype=np.int32)
            bst = np.zeros((G, nn + 1), dtype=np.int32)
            nng = np.zeros(G, dtype=np.int32)
            for i, (nodes, edges, eb, el, ci, ist) in enumerate(per):
                m, n_i = len(nodes), len(self.grids[i])
                Eb[i, :n_i], El[i, :n_i] = eb, el
                Cinv[i, :m * m] = ci.ravel()
                Cinv[i, m * m:2 * m * m] = np.linalg.inv(ci).ravel()
                istart[i, :len(ist)] = ist
                bst[i, :m + 1] = ccf_tables.bin_ranges(self.grids[i], edges)
                nng[i] = m
            one = (G == 1)
            T['Eb'] = torch.as_tensor(Eb[0] if one else Eb).to(dev)
            T['El'] = torch.as_tensor(El[0] if one else El).to(dev)
            T['Cinv'] = torch.as_tensor(Cinv[0] if one else Cinv).to(dev)
            T['istart'] = torch.as_tensor(istart[0] if one else istart).to(dev)
            T['bin_start'] = torch.as_tensor(bst[0] if one else bst).to(dev)
            T['nnode'] = nn
            if not one:
                T['nnode_g'] = torch.as_tensor(nng).to(dev)
        else:
            T['Eb'] = T['El'] = T['Cinv'] = T['istart'] = None
            T['nnode'], T['bin_start'] = 0, None
        if G > 1:
            T['npix_g'] = torch.as_tensor(self.npix_g).to(dev)


def make_resol(taps, nd, S, device):
    """device form of row taps [n, npix, nd] (n = 1: shared by all spectra)"""
    t = torch.as_tensor(np.ascontiguousarray(taps)).to(device)
    n, npix = t.shape[0], t.shape[1]
    unit = t.sum(dim=2)  # R @ 1
    if n == 1:
        unit = unit.expand(S, npix)
    return dict(taps=t.contiguous(), nd=int(nd),
                stride=0 if n == 1 else npix * int(nd),
                unit=unit.contiguous())


class SpecBatch:
    """A batch of S spectra, each observed in the same list of arms."""

    def __init__(self, arms):
        self.arms = list(arms)
        self.S = self.arms[0].S
        assert all(a.S == self.S for a in self.arms)
        self.device = self.arms[0].device
        self.names = [a.name for a in self.arms]
        self.badchi = 10 * sum(a.npix for a in self.arms)  # spec_fit.py:863
        # grid sets: a spectrum's own pixel count -> per-spectrum factor on badchi
        self.pen_scale = None
        if any(a.G > 1 for a in self.arms):
            n = sum(torch.as_tensor(a.npix_g.astype(np.float64)).to(self.device)[
                a.grid_id.long()] if a.G > 1 else
                torch.full((self.S, ), float(a.npix), dtype=torch.float64,
                           device=self.device) for a in self.arms)
            self.pen_scale = (10.0 * n / float(self.badchi)).contiguous()
        for a in self.arms:
            a.pen_scale = self.pen_scale

    @classmethod
    def from_specdata(cls, specdata_lists, device='cuda'):
        """specdata_lists: list (spectra) of lists (arms) of SpecData.  The
        spectra of an arm may share the wavelength grid (one grid, the fast
        path) or come on grids of their own (SDSS-style objects, spec_fit.py:
        70-145): the distinct grids of an arm become its grid set, spectra on
        shorter grids are padded (ArmData)."""
        first = specdata_lists[0]
        arms = []
        for ia, sd0 in enumerate(first):
            grids, index, gid = [], {}, []
            for sl in specdata_lists:
                lam = np.ascontiguousarray(sl[ia].lam, dtype=np.float64)
                key = (len(lam), lam.tobytes())
                if key not in index:
                    index[key] = len(grids)
                    grids.append(lam)
                gid.append(index[key])
            res = [getattr(sl[ia], 'resolution', None) for sl in specdata_lists]
            if any(r is not None for r in res) and any(r is None for r in res):
                raise ValueError('either every spectrum of an arm carries a '
                                 'resolution matrix or none does')
            if len(grids) == 1:
                arms.append(
                    ArmData(sd0.name, sd0.lam,
                            np.stack([sl[ia].spec for sl in specdata_lists]),
                            np.stack([sl[ia].espec for sl in specdata_lists]),
                            np.stack([np.asarray(sl[ia].badmask, dtype=np.uint8)
                                      for sl in specdata_lists]), device=device,
                            resolution=None if res[0] is None else
                            [r.mat for r in res]))
                continue
            npix = max(len(g) for g in grids)

            def padded(attr, fill, dtype):
                out = np.full((len(specdata_lists), npix), fill, dtype=dtype)
                for i, sl in enumerate(specdata_lists):
                    v = np.asarray(getattr(sl[ia], attr))
                    out[i, :len(v)] = v
                return out
            arms.append(ArmData(sd0.name, grids, padded('spec', 0.0, np.float64),
                                padded('espec', np.inf, np.float64),
                                padded('badmask', 1, np.uint8), device=device,
                                grid_id=np.array(gid, dtype=np.int32),
                                # (spec_fit.py:922-929: any SpecData.resolution --
                                # one matrix per spectrum, of its own grid's size)
                                resolution=None if res[0] is None else
                                [r.mat for r in res]))
        return cls(arms)

    def badchi_jobs(self, job_spec=None):
        """badchi of every job: the float of the batch, or (grid sets) a device
        tensor with the value of each job's spectrum"""
        if self.pen_scale is None:
            return float(self.badchi)
        ps = self.pen_scale if job_spec is None else self.pen_scale[job_spec.long()]
        return ps * float(self.badchi)

    def subset(self, idx):
        return SpecBatch([a.subset(idx) for a in self.arms])


def _chunks(n, size):
    for a in range(0, n, size):
        yield a, min(n, a + size)


# --------------------------------------------------------------------------
# template construction: A3/A4 -> A6 -> A7-construct
# --------------------------------------------------------------------------
def build_templates(lib, params, vsini=None, return_templ=False):
    """params [J, ndim] f64 device; vsini [J] f64 device or None.
    Returns coef [J, ntp, 4], outside [J] (+ the broadened template)."""
    L = _lib.lib()
    J = params.shape[0]
    templ, outside = lib.eval_batch(params)
    if vsini is not None:
        out = torch.empty_like(templ)
        vsini = vsini.to(torch.float64).contiguous()
        rc = L.rvs_vsini_convolve(_lib.ptr(templ), _lib.ptr(vsini),
                                  _lib.ptr(outside), lib.lnstep, 0.6, lib.ntp,
                                  J, _lib.ptr(out), _lib.stream())
        _lib.check(rc, 'rvs_vsini_convolve')
        templ = out
    coef = torch.empty((J, lib.ntp, 4), dtype=torch.float64, device=lib.device)
    # form 1: power-form records {y, b, c, d} consumed by the chi^2 kernels;
    # | 2: windowed solve, valid for the (log-)uniform grid of a library
    rc = L.rvs_spline_construct(_lib.ptr(lib.knots), _lib.ptr(templ), lib.ntp, J,
                                lib.spline_form, _lib.ptr(lib.spline_factors),
                                _lib.ptr(coef), _lib.stream())
    _lib.check(rc, 'rvs_spline_construct')
    if return_templ:
        return coef, outside, templ
    return coef, outside


def convolve_vsini(lib_or_lam, templ, vsini, eps=0.6):
    """templ [J, ntp] device, vsini [J] device"""
    L = _lib.lib()
    lam = lib_or_lam.lam if hasattr(lib_or_lam, 'lam') else lib_or_lam
    ratios = lam[1:] / lam[:-1]
    assert np.allclose(ratios, ratios[0]), "Wavelength grid must be logarithmic."
    lnstep = float(np.log(ratios[0]))
    out = torch.empty_like(templ)
    rc = L.rvs_vsini_convolve(_lib.ptr(templ.contiguous()), _lib.ptr(vsini),
                              None, lnstep, float(eps), templ.shape[1],
                              templ.shape[0], _lib.ptr(out), _lib.stream())
    _lib.check(rc, 'rvs_vsini_convolve')
    return out


# --------------------------------------------------------------------------
# chi^2 grid over velocities: A7-eval + A10 + A11 (+ penalties of A11)
# --------------------------------------------------------------------------
XCORR_PRUNE = True     # prune the last two FFT passes to the lags that are read
import os as _os
# rvs_chisq_grid's pack_min_jobs: 0 = library default (pack the Nv % 64
# left-over velocities of >= 2000 jobs), 1 = always, -1 = never (tests)
CG_PACK_MIN_JOBS = 0


def _arm_resol(arm, ia, resols):
    """resolution of arm ia: the `resol_params` override or the spectra's own"""
    if resols is not None and resols[ia] is not None:
        if arm.resol is not None:
            raise ValueError('You are not allowed to set resol_param together '
                             'with the resolution of each SpecData')
        return resols[ia]
    return arm.resol


def chisq_grid(batch, libs, coefs, outsides, vels, npoly=5, rbf=True,
               job_spec=None, job_templ=None, espec_sys=0.0,
               outside_penalty=True, out=None, vel_bounds=None, resols=None,
               defer_redo=False):
    """chi^2 of J jobs on a velocity grid, summed over the arms of `batch`.

    coefs[a]    [Tn, ntp_a, 4]   spline records of arm a
    outsides[a] [Tn]             outside flag of each template of arm a
    vels        [Nv] (shared) or [J, Nv]
    job_spec / job_templ int32 [J] or None (identity)
    Returns chisq [J, Nv], status int32 [J].
    """
    L = _lib.lib()
    dev = batch.device
    if job_spec is not None:
        J = job_spec.shape[0]
    elif job_templ is not None:
        J = job_templ.shape[0]
    else:
        J = batch.S
    vels = vels.to(device=dev, dtype=torch.float64).contiguous()
    shared = vels.dim() == 1
    Nv = vels.shape[-1]
    vstride = 0 if shared else Nv
    def by_point_kernel(rows):
        """every (job, velocity) of jobs `rows` (None = all) as a job of the
        point kernel: -> chisq [len(rows), Nv], status [len(rows)]"""
        js = job_spec if job_spec is not None else _arange32(0, J, dev)
        jt = job_templ if job_templ is not None else _arange32(0, J, dev)
        vv = vels[None, :].expand(J, Nv) if shared else vels
        if rows is not None:
            js, jt, vv = js[rows], jt[rows], vv[rows]
        n = js.shape[0]
        vv = vv.reshape(-1)
        res = torch.empty(n * Nv, dtype=torch.float64, device=dev)
        st = torch.zeros(n * Nv, dtype=torch.int32, device=dev)
        js2 = js.repeat_interleave(Nv).contiguous()
        jt2 = jt.repeat_interleave(Nv).contiguous()
        step = 1 << 18
        for a in range(0, n * Nv, step):
            b = min(n * Nv, a + step)
            res[a:b], st[a:b] = chisq_point(
                batch, libs, coefs, outsides, vv[a:b].contiguous(), npoly=npoly,
                rbf=rbf, job_spec=js2[a:b], job_templ=jt2[a:b],
                espec_sys=espec_sys, outside_penalty=outside_penalty,
                resols=resols)
        # third tier (spec_fit.py:337-354: Cholesky failed -> SVD): the point
        # kernel factors in-lane by Cholesky only, so what it could not factor
        # either goes through rvs_chisq_full, which carries the eigen branch
        bad = torch.nonzero(st & _lib.ST_CHOL_FALLBACK).reshape(-1)
        if bad.numel():
            full = chisq_full(batch, libs, coefs, vv[bad].contiguous(),
                              npoly=npoly, rbf=rbf, job_spec=js2[bad].contiguous(),
                              job_templ=jt2[bad].contiguous(), espec_sys=espec_sys,
                              want_models=False, resols=resols)
            tot = torch.zeros(bad.numel(), dtype=torch.float64, device=dev)
            stb = torch.zeros(bad.numel(), dtype=torch.int32, device=dev)
            for ia, f in enumerate(full):
                o = outsides[ia][jt2[bad].long()]
                bc = batch.badchi_jobs(js2[bad])
                pen = o * bc if outside_penalty else \
                    torch.where(torch.isfinite(o), torch.zeros_like(o), o)
                fin = torch.isfinite(pen)     # penalties of spec_fit.py:888-896
                tot += torch.where(fin, f['chisq'] + pen,
                                   torch.zeros_like(pen) + 1000.0 * bc)
                stb |= torch.where(fin, f['status'], torch.zeros_like(f['status']))
            res[bad] = tot
            stb = torch.where(torch.isfinite(tot), stb,
                              stb | _lib.ST_NONFINITE)
            st[bad] = stb | _lib.ST_CHOL_FALLBACK
        stj = torch.zeros(n, dtype=torch.int32, device=dev)
        st = st.reshape(n, Nv)
        for bit in range(12):
            stj |= ((st >> bit) & 1).amax(dim=1).to(torch.int32) << bit
        return res.reshape(n, Nv), stj

    wide = [_arm_resol(arm, ia, resols) for ia, arm in enumerate(batch.arms)]
    if npoly > POINT_MAXP or \
            any(r is not None and r['nd'] > RES_MAXND for r in wide):
        # resolution matrix wider than the grid kernel's band (the reference's
        # tests/test_sdss.py uses R = 50: 371 diagonals): every (job, velocity)
        # is a job of the point kernel, which applies a band of any width.
        # Likewise a continuum basis of 17 ... 32 functions (the reference has no
        # cap: spec_fit.py:860, :1018-1092; the grid kernel keeps at most 16 per
        # lane): chisq_point hands such jobs to rvs_chisq_full, a block per job
        res, stj = by_point_kernel(None)
        if out is not None:
            out.copy_(res)
            res = out
        return res, stj
    if out is None:
        out = torch.empty((J, Nv), dtype=torch.float64, device=dev)
    status = torch.zeros(J, dtype=torch.int32, device=dev)
    if vel_bounds is None:
        vel_bounds = (float(vels.min().item()), float(vels.max().item()))
    for ia, arm in enumerate(batch.arms):
        lib = libs[arm.name]
        work = arm.work(lib, espec_sys)
        polysT, logdet_off = arm.basis_ortho(npoly, rbf)
        o = outsides[ia]
        if job_templ is not None:
            o = o[job_templ.long()]
        pen = o * batch.badchi_jobs(job_spec) if outside_penalty else torch.where(
            torch.isfinite(o), torch.zeros_like(o), o)
        # + the constant log-determinant of the basis change (see basis_ortho);
        # a non finite penalty stays non finite (unusable template)
        if torch.is_tensor(logdet_off):   # grid set: the constant of each job's grid
            logdet_off = logdet_off if job_spec is None else \
                logdet_off[job_spec.long()]
        pen = (pen + logdet_off).contiguous()
        coef = coefs[ia]
        rs = _arm_resol(arm, ia, resols)
        for a, b in _chunks(J, 65535):
          with _ktime('chisq_grid', (b - a) * Nv):   # units: job-velocities
            js = _lib.ptr(job_spec[a:b]) if job_spec is not None else \
                (_lib.ptr(_arange32(a, b, dev)) if a > 0 else None)
            jt = _lib.ptr(job_templ[a:b]) if job_templ is not None else \
                (_lib.ptr(_arange32(a, b, dev)) if a > 0 else None)
            if rs is not None:   # A9: banded resolution matrix
                gid, G = arm.grid_args()
                rc = L.rvs_chisq_grid_resol_g(
                    _lib.ptr(arm.lam), _lib.ptr(polysT), _lib.ptr(work),
                    arm.npix, npoly, arm.S, gid, G, arm.basis_stride(npoly),
                    _lib.ptr(lib.knots), _lib.ptr(coef),
                    lib.ntp, coef.shape[0], int(lib.log_step),
                    _lib.ptr(rs['taps']), rs['nd'], rs['stride'], js, jt, b - a,
                    _lib.ptr(vels if shared else vels[a:b]), vstride, Nv,
                    _lib.ptr(pen[a:b]), float(batch.badchi),
                    0.0 if ia == 0 else 1.0, _lib.ptr(batch.pen_scale),
                    _lib.ptr(out[a:b]), _lib.ptr(status[a:b]), _lib.stream())
                _lib.check(rc, 'rvs_chisq_grid_resol')
                continue
            gid, G = arm.grid_args()
            rc = L.rvs_chisq_grid_g(
                _lib.ptr(arm.lam), _lib.ptr(polysT), _lib.ptr(work), arm.npix,
                npoly, arm.S, gid, G, arm.basis_stride(npoly),
                _lib.ptr(lib.knots), _lib.ptr(coef), lib.ntp,
                coef.shape[0], int(lib.log_step), js, jt, b - a,
                _lib.ptr(vels if shared else vels[a:b]), vstride, Nv,
                _lib.ptr(pen[a:b]), float(batch.badchi),
                0.0 if ia == 0 else 1.0, CG_PACK_MIN_JOBS,
                _lib.ptr(batch.pen_scale), _lib.ptr(out[a:b]),
                _lib.ptr(status[a:b]), _lib.stream())
            _lib.check(rc, 'rvs_chisq_grid')
    # Jobs whose normal matrix the velocity-grid kernel could not factor, or
    # found spanning > 1e9 in its pivots (RVS_ST_ILLCOND: a long stretch of
    # weightless pixels), are re-evaluated by the point kernel -- raw basis,
    # explicit residual, Cholesky -- and what that cannot factor either by
    # rvs_chisq_full's eigen branch (by_point_kernel: the reference's
    # Cholesky -> SVD tiers, spec_fit.py:337-354).  The look at the status
    # vector is a device-to-host synchronisation: at the end of the call, or --
    # defer_redo -- wherever the caller runs the returned finish().
    def finish():
        """the look at the status vector (host synchronisation) and the redo of
        the flagged jobs, in place; returns how many were redone"""
        redo = torch.nonzero(status & (_lib.ST_ILLCOND | _lib.ST_CHOL_FALLBACK)
                             ).reshape(-1)
        if redo.numel():
            res, stj = by_point_kernel(redo)
            out[redo] = res
            # (ST_ILLCOND stays on as the record that the job took this path)
            status[redo] = (status[redo] & ~(_lib.ST_NONFINITE |
                                             _lib.ST_CHOL_FALLBACK)) | stj | \
                _lib.ST_ILLCOND
        return int(redo.numel())
    if defer_redo:
        # the caller queues what does not depend on the redone jobs first and
        # calls finish() where it has to wait for the device anyway
        # (pipeline.fit_batch: at the end of the step)
        return out, status, finish
    finish()
    return out, status


# widest basis of rvs_chisq_point / rvs_chisq_grid (one lane keeps the packed
# normal matrix); rvs_chisq_full takes 32
POINT_MAXP = 16


def _per_arm(v, n):
    """scalar or per-arm sequence -> list of n floats"""
    if isinstance(v, (list, tuple)):
        assert len(v) == n
        return [float(_) for _ in v]
    return [float(v)] * n


def chisq_point(batch, libs, coefs, outsides, vel, npoly=5, rbf=True,
                job_spec=None, job_templ=None, espec_sys=0.0,
                outside_penalty=True, resols=None, fast_interp=False):
    """get_chisq for J (spectrum, template, velocity) triples, all arms in one
    launch set (rvs_chisq_point: lane per job, explicit residual norm).
    vel [J]; returns chisq [J], status int32 [J].  npoly 17 ... 32 (beyond the
    point kernel's 16 basis functions per lane): the arms' values from
    rvs_chisq_full, summed with the same penalties (spec_fit.py:888-896)."""
    import ctypes
    L = _lib.lib()
    dev = batch.device
    vel = vel.to(device=dev, dtype=torch.float64).contiguous()
    J = vel.shape[0]
    if npoly > POINT_MAXP:
        full = chisq_full(batch, libs, coefs, vel, npoly=npoly, rbf=rbf,
                          job_spec=job_spec, job_templ=job_templ,
                          espec_sys=espec_sys, want_models=False, resols=resols,
                          fast_interp=fast_interp)
        badchi = batch.badchi_jobs(job_spec)
        out = torch.zeros(J, dtype=torch.float64, device=dev)
        status = torch.zeros(J, dtype=torch.int32, device=dev)
        for f, o in zip(full, outsides):
            if job_templ is not None:
                o = o[job_templ.long()]
            usable = torch.isfinite(o)
            term = f['chisq'] + (o * badchi if outside_penalty else 0.0)
            out += torch.where(usable, term, 1000.0 * badchi + torch.zeros_like(o))
            status |= torch.where(usable, f['status'], torch.zeros_like(status))
        bad = ~torch.isfinite(out)
        status |= torch.where(bad, torch.full_like(status, _lib.ST_NONFINITE),
                              torch.zeros_like(status))
        return out, status
    narm = len(batch.arms)
    out = torch.empty(J, dtype=torch.float64, device=dev)
    status = torch.zeros(J, dtype=torch.int32, device=dev)
    nb = L.rvs_chisq_point_work_size(J, narm)
    scratch = torch.empty((nb + 7) // 8, dtype=torch.float64, device=dev)
    arr = (_lib.PointArm * narm)()
    keep = []
    esys = _per_arm(espec_sys, narm)
    for ia, arm in enumerate(batch.arms):
        lib = libs[arm.name]
        work = arm.work(lib, esys[ia])
        polysT = arm.basis(npoly, rbf)
        o = outsides[ia]
        if job_templ is not None:
            o = o[job_templ.long()]
        pen = o * batch.badchi_jobs(job_spec) if outside_penalty else torch.where(
            torch.isfinite(o), torch.zeros_like(o), o)
        pen = pen.contiguous()
        coef = coefs[ia]
        keep.append((work, polysT, pen, coef))
        a = arr[ia]
        a.lam, a.polysT = arm.lam.data_ptr(), polysT.data_ptr()
        a.spec, a.espec = arm.spec.data_ptr(), arm.espec.data_ptr()
        a.work, a.knots = work.data_ptr(), lib.knots.data_ptr()
        a.coef, a.penalty = coef.data_ptr(), pen.data_ptr()
        a.npix, a.S, a.ntp = arm.npix, arm.S, lib.ntp
        set_point_grid(a, arm, npoly)
        a.log_step = int(lib.log_step)
        a.espec_sys, a.fast_interp = esys[ia], int(bool(fast_interp))
        rs = _arm_resol(arm, ia, resols)
        if rs is not None:
            a.taps, a.taps_stride, a.nd = rs['taps'].data_ptr(), rs['stride'], \
                rs['nd']
    with _ktime('chisq_point', J):
        rc = L.rvs_chisq_point(ctypes.addressof(arr), narm, npoly,
                               _lib.ptr(job_spec), _lib.ptr(job_templ), J,
                               _lib.ptr(vel), float(batch.badchi),
                               _lib.ptr(scratch), _lib.ptr(out),
                               _lib.ptr(status), _lib.stream())
        _lib.check(rc, 'rvs_chisq_point')
    return out, status


def set_point_grid(p, arm, npoly):
    """grid-set fields of an rvs_point_arm (include/rvsgpu.h)"""
    p.G = arm.G
    p.grid_id = arm.grid_id.data_ptr() if arm.G > 1 else None
    p.polys_stride = arm.basis_stride(npoly)
    ps = getattr(arm, 'pen_scale', None)
    p.pen_scale = ps.data_ptr() if ps is not None else None


def fill_objective_arms(arr, batch, libs, npoly, rbf, espec_sys=0.0):
    """rvs_objective_arm descriptors of every arm (ctypes array `arr`); returns
    the tensors that must stay alive while the descriptors are used"""
    keep = []
    esys = _per_arm(espec_sys, len(batch.arms))
    for ia, arm in enumerate(batch.arms):
        lib = libs[arm.name]
        work = arm.work(lib, esys[ia])
        polysT = arm.basis(npoly, rbf)
        keep.append((work, polysT))
        a = arr[ia]
        p = a.pt
        p.lam, p.polysT = arm.lam.data_ptr(), polysT.data_ptr()
        p.spec, p.espec = arm.spec.data_ptr(), arm.espec.data_ptr()
        p.work, p.knots = work.data_ptr(), lib.knots.data_ptr()
        p.coef = p.penalty = p.taps = None
        p.taps_stride, p.espec_sys = 0, esys[ia]
        p.npix, p.S, p.ntp = arm.npix, arm.S, lib.ntp
        set_point_grid(p, arm, npoly)
        p.log_step, p.nd, p.fast_interp = int(lib.log_step), 0, 0
        a.factors = lib.spline_factors.data_ptr()
        a.lnstep = lib.lnstep
        a.ntp, a.ndim = lib.ntp, lib.ndim
        a.log_mask = lib.log_mask
        if lib.kind == 'regulargrid':
            a.dats, a.idgrid = lib.dats.data_ptr(), lib.idgrid.data_ptr()
            a.uvecs, a.vecs_s = lib.uvecs.data_ptr(), lib.vecs_s.data_ptr()
            a.ngrid = lib.ngrid
            for d in range(lib.ndim):
                a.ptp[d] = float(lib.ptp[d])
                a.lens[d] = int(lib.lens[d])
            a.exp_flag = lib.exp_flag
        else:   # rvs_objective_from_template: the grid fields are not read
            a.dats = a.idgrid = a.uvecs = a.vecs_s = None
            a.ngrid, a.exp_flag = 0, 0
    return keep


_max_ntp = {}


def can_fuse_objective(batch, libs, resols=None, fast_interp=False, npoly=10,
                       from_template=False):
    """the single-kernel objective needs a (log-)uniform template grid that fits
    LDS, and neither resolution matrices nor fast_interp; with the gather inside
    the kernel (rvs_objective_fused) regular-grid libraries, with the template
    handed over (from_template: rvs_objective_from_template) any evaluator"""
    if fast_interp or not FUSED_OBJECTIVE:
        return False
    if npoly not in _max_ntp:
        _max_ntp[npoly] = _lib.lib().rvs_objective_max_ntp(npoly)
    for ia, arm in enumerate(batch.arms):
        lib = libs[arm.name]
        if lib.spline_factors is None:
            return False
        if lib.kind != 'regulargrid' and not from_template:
            return False
        if lib.ntp > _max_ntp[npoly] or lib.ntp < 32:
            return False
        if npoly > 10 and (2 * arm.npix > lib.ntp or
                           2 * lib.ntp < 8 * (npoly * (npoly + 3) // 2 + 1)):
            return False   # (objective_kernel<P > 10>: csrc/objective.hip, RED_DYN)
        if _arm_resol(arm, ia, resols) is not None:
            return False
    return True


# False: the chain of stand-alone kernels also for regular-grid libraries
# (tests/test_gpu_parity.py::test_objective_fused compares the two)
FUSED_OBJECTIVE = True


def objective_fused(batch, libs, params, vsini, vel, npoly=5, rbf=True,
                    job_spec=None, espec_sys=0.0, outside_penalty=True,
                    njobs=None, out=None):
    """get_chisq for J (spectrum, parameters, vsini, velocity) jobs as ONE kernel
    per call (rvs_objective_fused): no template or spline record in HBM.
    Returns chisq [J], status int32 [J].  `njobs` (int32 device tensor, one
    element): only the first njobs[0] jobs are evaluated, `out` (given by the
    caller) keeps its values behind them (rvs_objective_fused_n)."""
    import ctypes
    L = _lib.lib()
    dev = batch.device
    vel = vel.to(device=dev, dtype=torch.float64).contiguous()
    params = params.to(device=dev, dtype=torch.float64).contiguous()
    J = vel.shape[0]
    narm = len(batch.arms)
    arr = (_lib.ObjectiveArm * narm)()
    keep = fill_objective_arms(arr, batch, libs, npoly, rbf, espec_sys)
    if out is None:
        out = torch.empty(J, dtype=torch.float64, device=dev)
    status = torch.zeros(J, dtype=torch.int32, device=dev)
    nb = L.rvs_objective_work_size(J, narm)
    scratch = torch.empty((nb + 7) // 8, dtype=torch.float64, device=dev)
    if vsini is not None:
        vsini = vsini.to(device=dev, dtype=torch.float64).contiguous()
    with _ktime('objective_fused', J):
        rc = L.rvs_objective_fused_n(ctypes.addressof(arr), narm, npoly,
                                     _lib.ptr(params), _lib.ptr(vsini),
                                     _lib.ptr(job_spec), J, _lib.ptr(njobs),
                                     _lib.ptr(vel), float(batch.badchi),
                                     int(outside_penalty), _lib.ptr(scratch),
                                     _lib.ptr(out), _lib.ptr(status),
                                     _lib.stream())
        _lib.check(rc, 'rvs_objective_fused')
    del keep
    return out, status


def objective_from_template(batch, libs, templs, outsides, vsini, vel, npoly=5,
                            rbf=True, job_spec=None, espec_sys=0.0,
                            outside_penalty=True):
    """objective_fused for evaluators that are no grid gather: templs[a]
    [J, ntp_a] (unbroadened), outsides[a] [J] (rvs_objective_from_template)."""
    import ctypes
    L = _lib.lib()
    dev = batch.device
    vel = vel.to(device=dev, dtype=torch.float64).contiguous()
    J = vel.shape[0]
    narm = len(batch.arms)
    arr = (_lib.ObjectiveArm * narm)()
    keep = fill_objective_arms(arr, batch, libs, npoly, rbf, espec_sys)
    out = torch.empty(J, dtype=torch.float64, device=dev)
    status = torch.zeros(J, dtype=torch.int32, device=dev)
    nb = L.rvs_objective_work_size(J, narm)
    scratch = torch.empty((nb + 7) // 8, dtype=torch.float64, device=dev)
    if vsini is not None:
        vsini = vsini.to(device=dev, dtype=torch.float64).contiguous()
    tp = (ctypes.c_void_p * narm)(*[t.data_ptr() for t in templs])
    op = (ctypes.c_void_p * narm)(*[o.data_ptr() for o in outsides])
    with _ktime('objective_from_template', J):
        rc = L.rvs_objective_from_template(
            ctypes.addressof(arr), narm, npoly, ctypes.cast(tp, ctypes.c_void_p),
            ctypes.cast(op, ctypes.c_void_p), _lib.ptr(vsini), _lib.ptr(job_spec),
            J, _lib.ptr(vel), float(batch.badchi), int(outside_penalty),
            _lib.ptr(scratch), _lib.ptr(out), _lib.ptr(status), _lib.stream())
        _lib.check(rc, 'rvs_objective_from_template')
    del keep
    return out, status


_ar_cache = {}


def _arange32(a, b, dev):
    key = (a, b, str(dev))
    if key not in _ar_cache:
        t = torch.arange(a, b, dtype=torch.int32, device=dev)
        # shared by later callers on other streams (vel_fit._process_split)
        torch.cuda.current_stream(t.device).synchronize()
        _ar_cache[key] = t
    return _ar_cache[key]


def grid_moments(chisq, vels, Np=1, nvel=None, quadratic=True):
    """chisq [G*Np, Nv] (jobs ordered group-major) -> res [G, 8], probs [G, Nv]"""
    L = _lib.lib()
    dev = chisq.device
    Nv = chisq.shape[-1]
    G = chisq.shape[0] // Np
    res = torch.empty((G, 8), dtype=torch.float64, device=dev)
    probs = torch.empty((G, Nv), dtype=torch.float64, device=dev)
    status = torch.zeros(G, dtype=torch.int32, device=dev)
    vels = vels.to(torch.float64).contiguous()
    vstride = 0 if vels.dim() == 1 else Nv
    rc = L.rvs_grid_moments(_lib.ptr(chisq.contiguous()), _lib.ptr(vels),
                            vstride, _lib.ptr(nvel), G, Np, Nv, int(quadratic),
                            _lib.ptr(res), _lib.ptr(probs), _lib.ptr(status),
                            _lib.stream())
    _lib.check(rc, 'rvs_grid_moments')
    return res, probs, status


def chisq_full(batch, libs, coefs, vel, npoly=5, rbf=True, job_spec=None,
               job_templ=None, espec_sys=0.0, unit_template=False,
               want_models=True, resols=None, fast_interp=False):
    """Per-arm full output (spec_fit.py:941-961) for one velocity per job."""
    L = _lib.lib()
    dev = batch.device
    J = batch.S if job_spec is None else job_spec.shape[0]
    res = []
    esys = _per_arm(espec_sys, len(batch.arms))
    for ia, arm in enumerate(batch.arms):
        lib = None if unit_template else libs[arm.name]
        polysT = arm.basis(npoly, rbf)
        chisq = torch.empty(J, dtype=torch.float64, device=dev)
        tchi = torch.empty(J, dtype=torch.float64, device=dev)
        coeffs = torch.empty((J, npoly), dtype=torch.float64, device=dev)
        ngood = torch.empty(J, dtype=torch.int32, device=dev)
        status = torch.zeros(J, dtype=torch.int32, device=dev)
        model = raw = None
        if want_models:
            model = torch.empty((J, arm.npix), dtype=torch.float64, device=dev)
            raw = torch.empty((J, arm.npix), dtype=torch.float64, device=dev)
        coef = None if unit_template else coefs[ia]
        rs = _arm_resol(arm, ia, resols)
        gid, G = arm.grid_args()
        rc = L.rvs_chisq_full_g(
            _lib.ptr(arm.lam), _lib.ptr(polysT), _lib.ptr(arm.spec),
            _lib.ptr(arm.espec), _lib.ptr(arm.badmask), arm.npix, npoly, arm.S,
            _lib.ptr(lib.knots) if lib else None, _lib.ptr(coef),
            lib.ntp if lib else 0, coef.shape[0] if coef is not None else 0,
            int(lib.log_step) if lib else 1, 1, int(unit_template),
            _lib.ptr(job_spec), _lib.ptr(job_templ), J,
            _lib.ptr(vel.contiguous()) if vel is not None else None,
            esys[ia], int(bool(fast_interp)),
            _lib.ptr(rs['taps']) if rs else None,
            rs['nd'] if rs else 0, rs['stride'] if rs else 0,
            _lib.ptr(chisq), _lib.ptr(coeffs),
            _lib.ptr(model), _lib.ptr(raw), _lib.ptr(tchi), _lib.ptr(ngood),
            _lib.ptr(status), gid, G, arm.basis_stride(npoly), _lib.stream())
        _lib.check(rc, 'rvs_chisq_full')
        res.append(dict(chisq=chisq, true_chisq=tchi, coeffs=coeffs,
                        ngood=ngood, status=status, model=model, raw_model=raw))
    return res


def chisq_continuum(batch, npoly=5, rbf=True):
    """get_chisq_continuum for a batch: per arm dict(true_chisq [S], ngood [S],
    status [S]) (rvs_chisq_continuum, one wave per spectrum).  Spectra whose
    normal matrix is numerically singular are redone by rvs_chisq_full, which
    carries the eigen (SVD) fallback of spec_fit.py:337-354."""
    L = _lib.lib()
    dev = batch.device
    res = []
    for arm in batch.arms:
        polysT = arm.basis(npoly, rbf)
        S = arm.S
        tchi = torch.empty(S, dtype=torch.float64, device=dev)
        ngood = torch.empty(S, dtype=torch.int32, device=dev)
        status = torch.zeros(S, dtype=torch.int32, device=dev)
        if npoly <= 16:
            ut = arm.resol['unit'] if arm.resol is not None else None
            gid, G = arm.grid_args()
            rc = L.rvs_chisq_continuum_g(_lib.ptr(polysT), _lib.ptr(arm.spec),
                                         _lib.ptr(arm.espec),
                                         _lib.ptr(arm.badmask),
                                         _lib.ptr(ut), arm.npix, npoly, S,
                                         None, None,
                                         _lib.ptr(tchi),
                                         _lib.ptr(ngood), _lib.ptr(status),
                                         gid, G, arm.basis_stride(npoly),
                                         _lib.stream())
            _lib.check(rc, 'rvs_chisq_continuum')
        else:
            status.fill_(_lib.ST_CHOL_FALLBACK)
        res.append(dict(true_chisq=tchi, ngood=ngood, status=status))
    return res


def chisq_continuum_fix(batch, res, npoly=5, rbf=True):
    """host-synchronising tail of chisq_continuum: redo flagged spectra with the
    two-tier kernel.  Returns res (updated in place)."""
    for ia, arm in enumerate(batch.arms):
        bad = torch.nonzero(res[ia]['status'] & _lib.ST_CHOL_FALLBACK).reshape(-1)
        if bad.numel() == 0:
            continue
        sub = SpecBatch([arm.subset(bad)])
        full = chisq_full(sub, None, None, None, npoly=npoly, rbf=rbf,
                          unit_template=True, want_models=False)[0]
        res[ia]['true_chisq'][bad] = full['true_chisq']
        res[ia]['ngood'][bad] = full['ngood']
    return res


# --------------------------------------------------------------------------
# CCF: A15 + A14
# --------------------------------------------------------------------------
def ccf_preprocess(arm, lib, config, details=False, maxerr=10.0):
    L = _lib.lib()
    T = arm.ccf_tables(lib, config)
    cc = lib.ccf_set(config)
    dev = arm.device
    nfft = cc['npoints']
    ps = torch.empty((arm.S, nfft), dtype=torch.float64, device=dev)
    pi = torch.empty((arm.S, nfft), dtype=torch.float64, device=dev)
    sse = torch.empty(arm.S, dtype=torch.float64, device=dev)
    status = torch.zeros(arm.S, dtype=torch.int32, device=dev)
    cont = pfit = None
    if details:
        cont = torch.empty((arm.S, arm.npix), dtype=torch.float64, device=dev)
        pfit = torch.zeros((arm.S, max(T['nnode'], 1)), dtype=torch.float64,
                           device=dev)
    with _ktime('ccf_preprocess', arm.S):
      rc = L.rvs_ccf_preprocess_g(
        _lib.ptr(arm.lam), _lib.ptr(arm.spec), _lib.ptr(arm.espec),
        _lib.ptr(arm.badmask), arm.npix, arm.S, int(cc['continuum']),
        _lib.ptr(T['Eb']), _lib.ptr(T['El']), _lib.ptr(T['Cinv']),
        _lib.ptr(T['istart']), T['nnode'], _lib.ptr(T['bin_start']),
        _lib.ptr(T['xind']), _lib.ptr(T['rw']), nfft, float(maxerr), _lib.ptr(ps),
        _lib.ptr(pi), _lib.ptr(sse), _lib.ptr(cont), _lib.ptr(pfit),
        _lib.ptr(status), arm.grid_args()[0], _lib.ptr(T['npix_g']),
        _lib.ptr(T['nnode_g']), _lib.stream())
    _lib.check(rc, 'rvs_ccf_preprocess')
    out = dict(proc_spec=ps, proc_ivar=pi, sse=sse, status=status)
    if details:
        out.update(cont=cont, pfit=pfit)
    return out


def ccf_fit(batch, libs, config, keep_all=False, max_chunk=None):
    """fitter_ccf.fit for a batch.  Returns a dict of device tensors:
    best_id [S] int64, best_vel [S], best_ccf [S, nvel], status [S],
    proc_spec / proc_ivar per arm."""
    L = _lib.lib()
    dev = batch.device
    S = batch.S
    ref = libs[batch.names[0]].ccf_set(config)
    Tn = ref['T']
    for n in batch.names[1:]:
        cc = libs[n].ccf_set(config)
        if cc['T'] != Tn:
            raise RuntimeError('CCF template counts are inconsistent across setups')
        if (not np.array_equal(ref['params'], cc['params'])
                or not np.array_equal(ref['vsinis'], cc['vsinis'],
                                      equal_nan=True)):
            raise RuntimeError('The parameters of the CCF templates do not match')
    tabs = [a.ccf_tables(libs[a.name], config) for a in batch.arms]
    nvel = len(tabs[0]['vgrid_host'])
    pre = [ccf_preprocess(a, libs[a.name], config) for a in batch.arms]
    sse = torch.stack([p['sse'] for p in pre]).contiguous()  # [narm, S]
    status = torch.zeros(S, dtype=torch.int32, device=dev)
    for p in pre:
        status |= p['status']
    # chunk spectra so that the [chunk, T, nvel] accumulator stays modest
    if max_chunk is None:
        max_chunk = max(1, min(S, int(2e9 // (Tn * nvel * 8))))
        nchunk = -(-S // max_chunk)
        max_chunk = -(-S // nchunk)  # equal-sized chunks
    res = torch.empty((S, 4), dtype=torch.float64, device=dev)
    best_ccf = torch.empty((S, nvel), dtype=torch.float64, device=dev)
    allchi = None
    for a, b in _chunks(S, max_chunk):
        n = b - a
        acc = torch.empty((n, Tn, nvel), dtype=torch.float64, device=dev)
        for ia, arm in enumerate(batch.arms):
            cc = libs[arm.name].ccf_set(config)
            T = tabs[ia]
            nfft = cc['npoints']
            work = torch.empty((n, 2, nfft // 2 + 1, 2), dtype=torch.float64,
                               device=dev)
            with _ktime('ccf_xcorr', n):
              rc = L.rvs_ccf_xcorr(
                _lib.ptr(pre[ia]['proc_spec'][a:b]),
                _lib.ptr(pre[ia]['proc_ivar'][a:b]), nfft, n,
                _lib.ptr(cc['fft']), _lib.ptr(cc['fft2']), Tn,
                _lib.ptr(T['twid']), int(cc['continuum']),
                _lib.ptr(T['lag_pos']), _lib.ptr(T['lag_vel']), T['nlag'],
                _lib.ptr(T['ilo']), _lib.ptr(T['vgrid']), nvel,
                0.0 if ia == 0 else 1.0, _lib.ptr(T['prune']), _lib.ptr(acc),
                _lib.ptr(work), _lib.stream())
            _lib.check(rc, 'rvs_ccf_xcorr')
        sse_c = sse[:, a:b].contiguous()
        rc = L.rvs_ccf_select(_lib.ptr(acc), _lib.ptr(sse_c), len(batch.arms),
                              n, Tn, _lib.ptr(tabs[0]['vgrid']), nvel,
                              _lib.ptr(res[a:b]), _lib.ptr(best_ccf[a:b]),
                              _lib.ptr(status[a:b]), _lib.stream())
        _lib.check(rc, 'rvs_ccf_select')
        if keep_all:
            allchi = acc + sse_c.sum(dim=0)[:, None, None]
    out = dict(best_id=res[:, 0].long(), best_vel=res[:, 1].contiguous(),
               best_pix=res[:, 2].long(), best_ccf=best_ccf, status=status,
               vel_grid=tabs[0]['vgrid'], vel_grid_host=tabs[0]['vgrid_host'],
               proc_spec=[p['proc_spec'] for p in pre],
               proc_ivar=[p['proc_ivar'] for p in pre],
               steps=[t['step'] for t in tabs])
    if keep_all:
        out['all_chisqs'] = allchi
    return out
