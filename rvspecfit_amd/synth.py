"""Synthetic stellar spectra and template libraries (numpy only).

This is the build's OWN generator.  It plays the role that
``tests/mktemps.py`` / ``tests/mktemps_grid.py`` play upstream (a fake
"PHOENIX" grid that needs no network data), but is a different model: a
black-body-like T^4/lambda continuum times a dense, deterministic list of
Gaussian absorption lines spread over 3400-10100 A, so that every DESI arm
(b/r/z) contains lines.  The module is deliberately free of torch / astropy so
that it runs both in the torch interpreter (bench, tests) and in the oracle
interpreter that imports the reference to make golden vectors.

Nothing here is on the hot path: it only manufactures inputs.
"""
import numpy as np

SPEED_OF_LIGHT = 299792.458  # km/s (scipy.constants value used by the reference)

PARNAMES = ('teff', 'logg', 'feh', 'alpha')


def line_table(lam_lo=3400.0, lam_hi=10100.0, spacing=19.0, seed=20260917):
    """Deterministic pseudo line list.

    Returns dict(cen, depth, kind, width) where kind 0 = metal line (strength
    follows feh), 1 = alpha-element line (follows feh+alpha), 2 = Balmer-like
    line (follows teff, broad)."""
    rng = np.random.RandomState(seed)
    n = int((lam_hi - lam_lo) / spacing)
    cen = lam_lo + (np.arange(n) + rng.uniform(0.15, 0.85, size=n)) * spacing
    depth = rng.uniform(0.08, 0.75, size=n)
    kind = rng.choice([0, 0, 0, 1, 1, 2], size=n)
    width = np.where(kind == 2, rng.uniform(1.5, 3.0, size=n),
                     rng.uniform(0.08, 0.25, size=n))
    return dict(cen=cen, depth=depth, kind=kind, width=width)


_LINES = line_table()


def continuum(lam, teff):
    """Smooth continuum, arbitrary flux units of order 1-100."""
    return (np.asarray(teff, dtype=float) / 5000.)**4 * (5000. / lam)


def spectrum(lam, teff, logg, feh, alpha, wresol=0.0, lines=None):
    """Synthetic spectrum on wavelengths `lam` (1-D) for scalar parameters.

    wresol is an extra Gaussian sigma (Angstrom) added in quadrature to every
    line, which mimics an instrumental LSF."""
    lines = _LINES if lines is None else lines
    lam = np.asarray(lam, dtype=np.float64)
    out = continuum(lam, teff)
    tnorm = (teff - 3000.) / 9000.
    press = 0.03 + 0.12 * logg  # pressure broadening grows with logg
    for cen, dep, kind, w0 in zip(lines['cen'], lines['depth'], lines['kind'],
                                  lines['width']):
        if cen < lam[0] - 40 or cen > lam[-1] + 40:
            continue
        wint = np.sqrt(w0**2 + press**2)
        w = np.sqrt(wint**2 + wresol**2)
        if kind == 0:
            amp = dep * 10**(0.45 * feh) * (1.15 - 0.7 * tnorm)
        elif kind == 1:
            amp = dep * 10**(0.45 * (feh + alpha)) * (1.1 - 0.6 * tnorm)
        else:
            amp = dep * (0.25 + 0.9 * tnorm) * (1.2 - 0.08 * logg)
        amp = np.clip(amp, 0, 0.92) * wint / w
        i0, i1 = np.searchsorted(lam, [cen - 7 * w, cen + 7 * w])
        if i1 > i0:
            x = (lam[i0:i1] - cen) / w
            out[i0:i1] *= 1 - amp * np.exp(-0.5 * x * x)
    return out


def template_lam_grid(lam_left, lam_right, step, deltav=1000.0):
    """Log-spaced template wavelength grid, same construction as the reference's
    template prep (make_interpol.py:313-323) so that pixel counts match the
    DESI recipe (6215 / 5303 / 6449)."""
    fac1 = 1 + deltav / SPEED_OF_LIGHT
    log_step_val = np.log(1 + step / (0.5 * (lam_left + lam_right)))
    return np.exp(
        np.arange(np.log(lam_left / fac1), np.log(lam_right * fac1),
                  log_step_val))


def regular_grid(nteff=7, nlogg=7, nfeh=7, nalpha=7, teff_range=(3000., 12000.),
                 logg_range=(0., 5.), feh_range=(-2., 0.), alpha_range=(0., 1.),
                 axes=None):
    """Returns (uvecs in PHYSICAL units, vec[4, N] physical) of a full regular
    grid, C-order over (teff, logg, feh, alpha).  `axes`: the four node vectors
    given explicitly (any lengths, any spacing -- the shape of a real PHOENIX
    library, whose dimensions have different lengths)."""
    if axes is not None:
        u = [np.asarray(a, dtype=np.float64) for a in axes]
    else:
        u = [np.linspace(teff_range[0], teff_range[1], nteff),
             np.linspace(logg_range[0], logg_range[1], nlogg),
             np.linspace(feh_range[0], feh_range[1], nfeh),
             np.linspace(alpha_range[0], alpha_range[1], nalpha)]
    G = np.meshgrid(*u, indexing='ij')
    vec = np.array([g.ravel() for g in G])
    return u, vec


def to_power_two(i):
    return 2**(int(np.ceil(np.log(i) / np.log(2))))


def make_interp_library(setup, lam_left, lam_right, step, grid_kw=None,
                        resol=None, holes=(), dtype=np.float32):
    """Build an in-memory polylinear library in the layout of the reference's
    `interp_%s.h5` + `interpdat_%s.npy` pair (make_nd.py:150-177):

    dats  float32 [N_grid, n_tpix]  log-flux
    vec   float64 [4, N_grid]       MAPPED parameters (log10 teff)
    uvecs list of 4 float64 arrays  unique mapped grid values
    idgrid int64 [n1,n2,n3,n4]      row of dats or -1 for holes
    lam   float64 [n_tpix]          log-spaced
    """
    grid_kw = grid_kw or {}
    u, vec = regular_grid(**grid_kw)
    lam = template_lam_grid(lam_left, lam_right, step)
    keep = np.ones(vec.shape[1], dtype=bool)
    for h in holes:
        keep[h] = False
    vec = vec[:, keep]
    if resol is None:
        wres = 0.0
    else:
        wres = 0.5 * (lam_left + lam_right) / resol / 2.35
    n = vec.shape[1]
    dats = np.empty((n, len(lam)), dtype=dtype)
    for i in range(n):
        dats[i] = np.log(spectrum(lam, *vec[:, i], wresol=wres))
    mvec = vec.copy()
    mvec[0] = np.log10(mvec[0])
    uv0 = [np.unique(mvec[i], return_inverse=True) for i in range(4)]
    uvecs = [_[0] for _ in uv0]
    idgrid = np.zeros([len(_) for _ in uvecs], dtype=np.int64) - 1
    idgrid[tuple(_[1] for _ in uv0)] = np.arange(n)
    return dict(setup=setup, lam=lam, dats=dats, vec=mvec, uvecs=uvecs,
                idgrid=idgrid, log_step=True, log_ids=(0,),
                parnames=PARNAMES, physical_vec=vec)


def fake_observation(lam, teff, logg, feh, alpha, vel, snr, rng,
                     wresol=0.0, slope=0.0):
    """Noisy observed spectrum at radial velocity vel (km/s), Gaussian noise
    with the given per-pixel S/N.  Returns spec, espec."""
    beta = vel / SPEED_OF_LIGHT
    lam_rest = lam * np.sqrt((1 - beta) / (1 + beta))
    spec0 = spectrum(lam_rest, teff, logg, feh, alpha, wresol=wresol)
    if slope:
        x = (lam - lam[0]) / (lam[-1] - lam[0])
        spec0 = spec0 * (1 + slope * (x - 0.5))
    espec = spec0 / snr
    spec = spec0 + espec * rng.standard_normal(len(lam))
    return spec, espec


DESI_ARMS = dict(
    b=dict(obs=(3600., 5800.1, 0.8), templ=(3500., 5900., 0.4)),
    r=dict(obs=(5760., 7620.1, 0.8), templ=(5660., 7720., 0.4)),
    z=dict(obs=(7520., 9824.1, 0.8), templ=(7420., 9924., 0.4)),
)


def write_fits_grid(prefix, wavefile, grid_kw=None, holes=(), lam_hr=None):
    """Write the synthetic grid as PHOENIX-like FITS files + a wavelength file
    so that the REFERENCE's own prep pipeline (read_grid -> make_interpol ->
    make_nd -> make_ccf) can ingest it.  Only used by tests/golden/make_golden.py
    (needs astropy, i.e. the oracle interpreter)."""
    import os
    import astropy.io.fits as pyfits
    u, vec = regular_grid(**(grid_kw or {}))
    os.makedirs(prefix + '/specs', exist_ok=True)
    k = 0
    for i in range(vec.shape[1]):
        if i in holes:
            continue
        teff, logg, feh, alpha = vec[:, i]
        sp = spectrum(lam_hr, teff, logg, feh, alpha)
        hdr = pyfits.Header(dict(PHXTEFF=teff, PHXLOGG=logg, PHXM_H=feh,
                                 PHXALPHA=alpha))
        pyfits.writeto(prefix + '/specs/syn_%05d.fits' % k, sp, hdr,
                       overwrite=True)
        k += 1
    pyfits.writeto(prefix + '/' + wavefile, lam_hr, overwrite=True)
    return vec


# --------------------------------------------------------------------------
# vectorised generators (bench / large parity cases).  Same model as
# spectrum(); `xp` is numpy or torch so the observed spectra of a 10 000 spectra
# batch can be synthesised directly in HBM.
# --------------------------------------------------------------------------
def spectra_batch(lam, teff, logg, feh, alpha, vel=None, wresol=0.0, xp=np,
                  lines=None):
    """[B, npix] spectra for parameter vectors (arrays of length B) observed at
    radial velocities `vel` (km/s) on the common grid `lam` (numpy, host).
    With xp=torch the parameter arrays are device tensors and so is the result."""
    lines = _LINES if lines is None else lines
    lam_np = np.asarray(lam, dtype=np.float64)
    if xp is np:
        lamx = lam_np
        f = 1.0
        if vel is not None:
            beta = np.asarray(vel) / SPEED_OF_LIGHT
            f = np.sqrt((1 - beta) / (1 + beta))[:, None]
        clip = np.clip
    else:
        lamx = xp.as_tensor(lam_np, device=teff.device)
        f = 1.0
        if vel is not None:
            beta = vel / SPEED_OF_LIGHT
            f = xp.sqrt((1 - beta) / (1 + beta))[:, None]
        clip = xp.clamp
    lr = lamx[None, :] * f  # rest-frame wavelengths [B, npix]
    out = (teff[:, None] / 5000.)**4 * (5000. / lr)
    tnorm = (teff - 3000.) / 9000.
    press = 0.03 + 0.12 * logg
    vmax = 0.0 if vel is None else 0.0045
    for cen, dep, kind, w0 in zip(lines['cen'], lines['depth'], lines['kind'],
                                  lines['width']):
        if cen < lam_np[0] * (1 - vmax) - 40 or cen > lam_np[-1] * (1 + vmax) + 40:
            continue
        wint = xp.sqrt(w0**2 + press**2)
        w = xp.sqrt(wint**2 + wresol**2)
        if kind == 0:
            amp = dep * 10**(0.45 * feh) * (1.15 - 0.7 * tnorm)
        elif kind == 1:
            amp = dep * 10**(0.45 * (feh + alpha)) * (1.1 - 0.6 * tnorm)
        else:
            amp = dep * (0.25 + 0.9 * tnorm) * (1.2 - 0.08 * logg)
        amp = clip(amp, 0, 0.92) * wint / w
        wmax = float(w.max())
        i0, i1 = np.searchsorted(lam_np, [cen * (1 - vmax) - 7 * wmax,
                                          cen * (1 + vmax) + 7 * wmax])
        if i1 > i0:
            x = (lr[:, i0:i1] - cen) / w[:, None]
            out[:, i0:i1] *= 1 - amp[:, None] * xp.exp(-0.5 * x * x)
    return out


def make_interp_library_fast(setup, lam_left, lam_right, step, grid_kw=None,
                             resol=None, dtype=np.float32, device=None,
                             chunk=2048):
    """Vectorised make_interp_library (no holes).  With `device` (a torch
    device) the rows are synthesised there, `chunk` grid points at a time, and
    `dats` is a float32 device tensor -- how the bench builds libraries of
    realistic size (10^4 templates, hundreds of MB per arm) in seconds."""
    grid_kw = grid_kw or {}
    u, vec = regular_grid(**grid_kw)
    lam = template_lam_grid(lam_left, lam_right, step)
    wres = 0.0 if resol is None else 0.5 * (lam_left + lam_right) / resol / 2.35
    if device is None:
        sp = spectra_batch(lam, vec[0], vec[1], vec[2], vec[3], wresol=wres)
        dats = np.log(sp).astype(dtype)
    else:
        import torch
        n = vec.shape[1]
        dats = torch.empty((n, len(lam)), dtype=torch.float32, device=device)
        for i0 in range(0, n, chunk):
            v = [torch.as_tensor(vec[k, i0:i0 + chunk]).to(device)
                 for k in range(4)]
            sp = spectra_batch(lam, v[0], v[1], v[2], v[3], wresol=wres,
                               xp=torch)
            dats[i0:i0 + chunk] = torch.log(sp).float()
    mvec = vec.copy()
    mvec[0] = np.log10(mvec[0])
    uv0 = [np.unique(mvec[i], return_inverse=True) for i in range(4)]
    uvecs = [_[0] for _ in uv0]
    idgrid = np.zeros([len(_) for _ in uvecs], dtype=np.int64) - 1
    idgrid[tuple(_[1] for _ in uv0)] = np.arange(vec.shape[1])
    return dict(setup=setup, lam=lam, dats=dats, vec=mvec, uvecs=uvecs,
                idgrid=idgrid, log_step=True, log_ids=(0, ),
                parnames=PARNAMES, physical_vec=vec)


def library_as_npz_dict(lib, ccf=None):
    """Flatten to the converted-artefact key set (tests/golden/lib_*.npz) that
    both library.TemplateLibrary and the oracle's Library read."""
    d = dict(lam=lib['lam'], dats=lib['dats'], vec=lib['vec'],
             idgrid=lib['idgrid'], log_step=np.array(True),
             log_ids=np.array(lib['log_ids'], dtype=np.int64),
             parnames=np.array(list(lib['parnames'])))
    for i, u in enumerate(lib['uvecs']):
        d['uvec%d' % i] = u
    if ccf is not None:
        d.update(ccf)
    return d


def make_ccf_templates(lib, lam0, lam1, step, every=64, vsinis=(0., 300.),
                       convolve=None, cont=None):
    """CCF template set for a synthetic library in the reference's artefact
    layout (make_ccf.py:417-493): continuum-normalised, optionally rotationally
    broadened models rebinned on exp(linspace(log lam0, log lam1, 2^k)), their
    rfft and the rfft of their squares.  The continuum of the synthetic model is
    known analytically, so no robust fit is needed here (offline prep, not part
    of the hot path).  `convolve(lam, templ[J, n], vsini[J])` is the vsini
    broadening routine (the HIP kernel in the bench).  `cont`: the models'
    continuum where it is not the synthetic family's (a float or [n_sel, ntp])."""
    npoints = to_power_two(int((lam1 - lam0) / step))
    logl = np.linspace(np.log(lam0), np.log(lam1), npoints)
    sel = np.arange(0, lib['dats'].shape[0], every)
    phys = lib['physical_vec'][:, sel]
    models, params, vs_list = [], [], []
    rows = lib['dats'][sel]
    if not isinstance(rows, np.ndarray):     # device tensor (bench, big grids)
        rows = rows.cpu().numpy()
    flux = np.exp(rows.astype(np.float64))
    if cont is None:
        cont = continuum(lib['lam'][None, :], phys[0][:, None])
    else:
        cont = np.broadcast_to(np.asarray(cont, dtype=np.float64), flux.shape)
    for vs in vsinis:
        m = flux
        if vs and vs > 0:
            m = convolve(lib['lam'], flux, np.full(len(sel), float(vs)))
        for i in range(len(sel)):
            c = np.interp(logl, np.log(lib['lam']), m[i] / cont[i], left=1.,
                          right=1.)
            models.append(c)
            params.append(phys[:, i])
            vs_list.append(vs)
    # reference order: for model: for vsini (make_ccf.py:263-272)
    order = np.argsort(np.tile(np.arange(len(sel)), len(vsinis)), kind='stable')
    models = np.array(models)[order]
    params = np.array(params)[order]
    vs_list = np.array(vs_list, dtype=float)[order]
    splinestep = max(1000., 3e5 * (np.exp((logl[-1] - logl[0]) / 20) - 1))
    return dict(ccf_fft=np.fft.rfft(models, axis=1),
                ccf_fft2=np.fft.rfft(models**2, axis=1), ccf_mod=models,
                ccf_params=params, ccf_vsinis=vs_list,
                ccf_parnames=np.array(list(PARNAMES)),
                ccf_logl0=np.array(logl[0]), ccf_logl1=np.array(logl[-1]),
                ccf_npoints=np.array(npoints), ccf_continuum=np.array(True),
                ccf_splinestep=np.array(splinestep),
                ccf_maxcontpts=np.array(20))
