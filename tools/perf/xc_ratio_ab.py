"""rvs_ccf_xcorr in the mode without continuum normalisation (-c0^2 / c1,
fitter_ccf.py:204-207): the wave-specialised persistent kernel against the per-pair
kernel (option xc_ws = 0), B spectra x T templates at nfft 8192 / 4096, and the
continuum mode beside them.  usage: xc_ratio_ab.py [B] [T]"""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, '.')
from rvspecfit_amd import _lib, ccf_tables

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 76
L = _lib.lib()
for nfft in (8192, 4096):
    n2 = nfft // 2
    rng = np.random.RandomState(1)
    spec = 1 + 0.2 * rng.standard_normal((B, nfft))
    ivar = rng.uniform(0.5, 2.0, (B, nfft))
    tmod = 1 + 0.3 * rng.standard_normal((T, nfft))
    tfft, tfft2 = np.fft.rfft(tmod, axis=1), np.fft.rfft(tmod**2, axis=1)
    step, nl = 10.0, 200
    maxvel = (nl - 0.5) * step
    off = nfft // 2
    vels = -((np.arange(nfft) + off) % nfft - off) * step
    sel = np.abs(vels) < (maxvel + step)
    ind = np.roll(np.nonzero(sel)[0], sel.sum() // 2)[::-1]
    sub = np.ascontiguousarray(vels[ind])
    vgrid = np.linspace(-maxvel, maxvel, 400)
    ilo = ccf_tables.interp_tables(sub, vgrid)
    pos = np.array([L.rvs_ccf_fft_pos(nfft, int(n) >> 1) for n in ind])
    lag_pos = (2 * pos + (ind & 1)).astype(np.int32)
    prune = None
    l2 = n2.bit_length() - 1
    if l2 % 3 == 0 and l2 >= 6:
        pm = np.zeros(n2 // 64 + n2 // 8, dtype=np.uint8)
        for p_ in pos:
            pm[n2 // 64 + (int(p_) >> 3)] |= 1 << (int(p_) & 7)
            pm[int(p_) >> 6] |= 1 << ((int(p_) >> 3) & 7)
        prune = torch.as_tensor(pm).to('cuda')
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to('cuda')
    twid = np.exp(2j * np.pi * np.arange(n2) / nfft)
    t_spec, t_ivar, t_f, t_f2 = d(spec), d(ivar), d(tfft), d(tfft2)
    t_tw, t_lp, t_lv, t_ilo, t_vg = d(twid), d(lag_pos), d(sub), d(ilo), d(vgrid)
    out = torch.zeros((B, T, len(vgrid)), dtype=torch.float64, device='cuda')
    work = torch.empty((B, 2, n2 + 1), dtype=torch.complex128, device='cuda')

    def call(cont):
        rc = L.rvs_ccf_xcorr(_lib.ptr(t_spec), _lib.ptr(t_ivar), nfft, B,
                             _lib.ptr(t_f), _lib.ptr(t_f2), T, _lib.ptr(t_tw), cont,
                             _lib.ptr(t_lp), _lib.ptr(t_lv), len(sub), _lib.ptr(t_ilo),
                             _lib.ptr(t_vg), len(vgrid), 0.0, _lib.ptr(prune),
                             _lib.ptr(out), _lib.ptr(work), _lib.stream())
        assert rc == 0

    for cont in (1, 0):
        res = {}
        for ws in (1, 0, 1, 0):
            with _lib.option('xc_ws', ws):
                call(cont)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    call(cont)
                torch.cuda.synchronize()
                res.setdefault(ws, []).append((time.perf_counter() - t0) / 3 * 1e3)
        print('nfft %d B %d T %d nlag %d %s: persistent %s ms, per pair %s ms' % (
            nfft, B, T, len(sub), 'continuum' if cont else '-c0^2/c1',
            ['%.2f' % v for v in res[1]], ['%.2f' % v for v in res[0]]))
