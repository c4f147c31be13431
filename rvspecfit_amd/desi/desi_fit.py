"""DESI survey driver: ingestion, conditioning, warning bits and the rvtab/rvmod
products -- mirror of py/rvspecfit/desi/desi_fit.py, with the per-fibre
`poolex.submit(proc_onespec, ...)` loop (desi_fit.py:1176-1217) replaced by ONE
batched call per file: every selected fibre of a coadd/spectra file is
conditioned on the host, uploaded once, and fitted in lock-step on the GPU
(fitter_ccf.fit -> vel_fit.process -> spec_fit.get_chisq_continuum on a
SpecBatch).

What is restated here (reference lines in each docstring):
  bitmasks, get_rvs_warn            the RVS_WARN bits
  get_sns, fiberstatus_select,
  select_fibers_to_fit              fibre selection
  interpolate_bad_regions,
  get_specdata(_batch)              masking / sigma-clamping rules
  resolution_mat_*, deconvolve_*,
  construct_resolution_sparse_matrix  DESI resolution data -> banded matrix
  get_column_desc, get_prim_header,
  proc_onespec, proc_desi, proc_desi_wrapper, proc_many
                                    the RVTAB / RVMOD schema and files
  main                              the command line, with the reference's option names

File I/O goes through fits_min (astropy is not part of the image).  Outside
the path and not built: the MPI file server behind --mpi (one process drives one
GPU; ranks of torch.distributed.run stride the file list or share a queue file), plots (`make_plot`;
doplot is accepted and ignored with a warning), the desitarget object-type
filter (the reference ignores `objtypes` itself when desitarget is missing,
desi_fit.py:584-587, 617-623 -- so does this module, always).

Arithmetic note: the flux/ivar images are float32 and the reference conditions
them with numpy expressions whose result types depend on numpy's scalar
promotion rules.  The golden vectors were produced with numpy 1.26 (legacy
value-based casting); the casts are written out explicitly below so that the
result is the same under numpy 2.
"""
import enum
import logging
import os
import sys
import time
import warnings

import numpy as np
import scipy.linalg
import scipy.sparse

from .. import fits_min as pyfits
from .. import spec_fit, spec_inter, utils


class ProcessStatus(enum.Enum):
    SUCCESS = 0
    FAILURE = 1
    EXISTING = 2

    def __str__(self):
        return self.name


class GlobalConfig:
    table_prefix = 'rvtab'
    model_prefix = 'rvmod'


DEPEND_PACKAGES = ['numpy', 'scipy', 'torch', 'pyyaml']

# desi_fit.py:50-58
bitmasks = {
    'CHISQ_WARN': 1,  # delta chi-square vs continuum is too small
    'RV_WARN': 2,  # rv is too close to the edge
    'RVERR_WARN': 4,  # RV error is too large
    'PARAM_WARN': 8,  # parameters are too close to the edge
    'VSINI_WARN': 16,  # vsini is too large
    'BAD_SPECTRUM': 32,  # some issue with the spectrum
    'BAD_HESSIAN': 64  # issue with the hessian matrix
}

# units the reference attaches through astropy.units (desi_fit.py:312-330, 353)
COLUMN_UNITS = {
    'VRAD': 'km s-1', 'VRAD_ERR': 'km s-1', 'VSINI': 'km s-1',
    'VRAD_CCF': 'km s-1', 'TEFF': 'K', 'TEFF_ERR': 'K'
}
INT_NULL = 999999  # astropy's fill value for masked integer cells


def update_process_status_file(status_fname, processed_file, status, nobjects,
                               time_sec, start=False):
    """desi_fit.py:61-74"""
    if start:
        with open(status_fname, 'w'):
            pass
        if processed_file is None:
            return
    with open(status_fname, 'a') as fp:
        print(f'{processed_file} {status} {nobjects} {time_sec:.2f}', file=fp)


def get_dep_versions():
    """desi_fit.py:77-90"""
    from importlib.metadata import version, PackageNotFoundError
    ret = {}
    for curp in DEPEND_PACKAGES:
        try:
            ret[curp] = version(curp)
        except (ImportError, PackageNotFoundError):
            pass
    from .. import __version__ as v
    ret['rvspecfit_amd'] = v
    ret['python'] = str.split(sys.version, ' ')[0]
    return ret


def get_zbest_fname(fname):
    """desi_fit.py:93-116: the redrock/zbest file next to a coadd-/spectra- file"""
    paths = fname.split('/')
    fname_end = paths[-1]
    if fname_end[-3:] == '.gz':
        fname_end = fname_end[:-3]
    not_found = (None, None)
    for curpref in ('coadd-', 'spectra-'):
        if fname_end[:len(curpref)] == curpref:
            break
    else:
        return not_found
    for cur_zpref, cur_ext in zip(('redrock-', 'zbest-'),
                                  ('REDSHIFTS', 'ZBEST')):
        f1 = fname_end.replace(curpref, cur_zpref)
        for postf in ('', '.gz'):
            zbest_path = '/'.join(paths[:-1] + [f1]) + postf
            if os.path.exists(zbest_path):
                return zbest_path, cur_ext
    return not_found


def get_prim_header(versions=None, config=None, cmdline=None,
                    spectrum_header=None, zbest_path=None):
    """desi_fit.py:119-156"""
    header = pyfits.Header()
    for i, (k, v) in enumerate(get_dep_versions().items()):
        header['DEPNAM%02d' % i] = (k, 'Software')
        header['DEPVER%02d' % i] = (v, 'Version')
    for i, (k, v) in enumerate((versions or {}).items()):
        header['TMPLCON%d' % i] = (k, 'Spec arm config name')
        header['TMPLREV%d' % i] = (v['revision'], 'Spec template revision')
        header['TMPLSVR%d' % i] = (v['creation_soft_version'],
                                   'Spec template soft version')
    if config is not None:
        header['RVS_CONF'] = config['config_file_path']
    if cmdline is not None:
        header['RVS_CMD'] = cmdline
    header['RR_FILE'] = (zbest_path or '', 'Redrock redshift file')
    copy_keys = [
        'SPGRP', 'SPGRPVAL', 'TILEID', 'SPECTRO', 'PETAL', 'NIGHT', 'EXPID',
        'HPXPIXEL', 'HPXNSIDE', 'HPXNEST'
    ]
    if spectrum_header is not None:
        for key in copy_keys:
            if key in spectrum_header:
                header[key] = spectrum_header[key]
    return header


def valid_file(FP):
    """desi_fit.py:225-245"""
    extnames = [_.name for _ in FP]
    reqnames = ['%s_%s' % (a, p) for a in ('B', 'R', 'Z')
                for p in ('WAVELENGTH', 'FLUX', 'IVAR', 'MASK')] + ['FIBERMAP']
    missing = [_ for _ in reqnames if _ not in extnames]
    if missing:
        logging.warning('Extensions %s are missing' % (','.join(missing)))
        return False
    return True


# ------------------------------------------------------------- warning bits
def _bad_edge_check(value, edges, threshold):
    """desi_fit.py:433-441 (elementwise)"""
    return (value < edges[0] + threshold) | (value > edges[1] - threshold)


def rvs_warn_bits(chisq_tot, chisq_c_tot, vrad, vsini, vrad_err, bad_hessian,
                  teff, feh, logg, config):
    """get_rvs_warn (desi_fit.py:381-430) on arrays: int64 [n]."""
    chisq_tot = np.asarray(chisq_tot, dtype=np.float64)
    w = np.zeros(chisq_tot.shape, dtype=np.int64)
    dchisq = np.asarray(chisq_c_tot, dtype=np.float64) - chisq_tot
    w[dchisq < 50] |= bitmasks['CHISQ_WARN']
    w[_bad_edge_check(np.asarray(vrad), [config['min_vel'], config['max_vel']],
                      5)] |= bitmasks['RV_WARN']
    w[np.asarray(vsini) > 100] |= bitmasks['VSINI_WARN']
    w[np.asarray(vrad_err) > 100] |= bitmasks['RVERR_WARN']
    w[np.asarray(bad_hessian, dtype=bool)] |= bitmasks['BAD_HESSIAN']
    for val, edges, thresh in ((teff, [2300, 15000], 10), (feh, [-4, 1], 0.01),
                               (logg, [-.5, 6.5], 0.01)):
        w[_bad_edge_check(np.asarray(val), edges, thresh)] |= \
            bitmasks['PARAM_WARN']
    return w


def get_rvs_warn(fit_res, outdict, config):
    """desi_fit.py:381-430 for one fit (plain floats instead of Quantities)."""
    p = fit_res['param']
    return int(rvs_warn_bits([outdict['CHISQ_TOT']], [outdict['CHISQ_C_TOT']],
                             [outdict['VRAD']], [outdict['VSINI']],
                             [outdict['VRAD_ERR']], [fit_res['bad_hessian']],
                             [p['teff']], [p['feh']], [p['logg']], config)[0])


# --------------------------------------------------------- fibre selection
def get_sns(data, ivars, masks):
    """desi_fit.py:444-456: vector of per-fibre median S/N"""
    usable = (ivars > 0) & ~(masks > 0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')   # (rows without a usable pixel)
        sn = np.where(usable, data * np.sqrt(np.where(usable, ivars, 0.0)), np.nan)
        med = np.nanmedian(sn, axis=1)
    return np.where(np.isfinite(med), med, -1e9)


def read_data(FP, setups):
    """desi_fit.py:459-492"""
    fluxes, ivars, waves, masks, resolutions = {}, {}, {}, {}, {}
    for s in setups:
        S = s.upper()
        fluxes[s] = FP['%s_FLUX' % S].data
        ivars[s] = FP['%s_IVAR' % S].data
        masks[s] = FP['%s_MASK' % S].data
        waves[s] = FP['%s_WAVELENGTH' % S].data
        resolutions[s] = FP['%s_RESOLUTION' % S].data \
            if ('%s_RESOLUTION' % S) in FP else None
    return fluxes, ivars, masks, waves, resolutions


def fiberstatus_select(fibermap):
    """desi_fit.py:524-543: only RESTRICTED (3) | VARIABLE (20) bits allowed"""
    good_fiberstatus = int(np.sum(1 << np.array([3, 20], dtype=int)))
    names = fibermap.columns.names
    if 'FIBERSTATUS' in names:
        col = fibermap['FIBERSTATUS']
    elif 'COADD_FIBERSTATUS' in names:
        col = fibermap['COADD_FIBERSTATUS']
    else:
        raise Exception('Fiberstatus column not found')
    return (col & good_fiberstatus) == col


def _arm_sns(scores, setups, fluxes, ivars, masks):
    """Per-arm median S/N of every row: the pipeline's own SCORES column when the
    file has one (in the reference's order of preference, desi_fit.py:1049-1067),
    else computed from the pixels (get_sns)."""
    have = scores.columns.names
    for stem in ('MEDIAN_CALIB_SNR_', 'MEDIAN_COADD_SNR_',
                 'MEDIAN_COADD_FLUX_SNR_'):
        if stem + setups[0].upper() in have:
            return {a: scores[stem + a.upper()] for a in setups}
    return {a: get_sns(fluxes[a], ivars[a], masks[a]) for a in setups}


def _redrock_rows(fibermap, zbest_path, zbest_ext, zbest_type='STAR',
                  zbest_maxvel=1500):
    """The redshift file of a coadd, aligned with the fibermap rows
    (desi_fit.py:639-672): returns (stellar mask [n], Z, SPECTYPE, SUBTYPE).  A
    file with one row per fibre is taken as is (TARGETIDs must agree); any
    other is matched by TARGETID -- fibres without a row get NaN / empty and
    are not stellar."""
    zb = pyfits.open(zbest_path)[zbest_ext].data
    z, spectype, subtype = zb['Z'], zb['SPECTYPE'], zb['SUBTYPE']
    stellar = (spectype == zbest_type) | (np.abs(z) < zbest_maxvel / 3e5)
    tid = fibermap['TARGETID']
    if len(zb) == len(tid):
        assert np.all(zb['TARGETID'] == tid)
        return stellar, z, spectype, subtype
    row_of = {int(t): i for i, t in enumerate(zb['TARGETID'])}
    pos = np.array([row_of.get(int(t), -1) for t in tid])
    hit = pos >= 0
    cols = []
    for src, fill in ((z, np.nan), (spectype, None), (subtype, None)):
        dst = np.zeros(len(tid), dtype=src.dtype)
        if fill is not None:
            dst = dst + fill
        dst[hit] = src[pos[hit]]
        cols.append(dst)
    return np.isin(tid, zb['TARGETID'][stellar]), cols[0], cols[1], cols[2]


def select_fibers_to_fit(fibermap, sns, zbest_path=None, zbest_ext=None,
                         minsn=None, objtypes=None, expid_range=None,
                         fit_targetid=None, zbest_select=False,
                         zbest_include=False):
    """Which rows of a file are fitted (the rules of desi_fit.py:546-679, one
    boolean column per rule, all of them ANDed): exposure window, usable
    FIBERSTATUS, not SKY / BAD, requested TARGETIDs, best-arm S/N above `minsn`,
    and -- with `zbest_select` -- stellar according to redrock.  `objtypes`
    needs desitarget, which the reference treats as optional too: never applied
    here.  Returns (subset, rr_z, rr_spectype, rr_subtype); the redrock columns
    are None unless a redshift file was read."""
    n = len(fibermap)
    rules = [fiberstatus_select(fibermap),
             (fibermap['OBJTYPE'] != 'SKY') & (fibermap['OBJTYPE'] != 'BAD')]
    if 'EXPID' in fibermap.columns.names:
        lo, hi = (None, None) if expid_range is None else expid_range
        lo = -1 if lo is None else lo
        hi = np.inf if hi is None else hi
        rules.append((fibermap['EXPID'] > lo) & (fibermap['EXPID'] <= hi))
    if fit_targetid is not None:
        rules.append(np.isin(fibermap['TARGETID'], fit_targetid))
    if minsn is not None:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            rules.append(np.max(np.array(list(sns.values())), axis=0) > minsn)
    rr_z = rr_spectype = rr_subtype = None
    if zbest_select or zbest_include:
        if zbest_path is None:
            logging.warning(
                'zbest selection requested, but the zbest file not found')
        else:
            logging.info('Using redshift file %s', zbest_path)
            stellar, rr_z, rr_spectype, rr_subtype = _redrock_rows(
                fibermap, zbest_path, zbest_ext)
            if zbest_select:
                rules.append(stellar)
    subset = np.ones(n, dtype=bool)
    for r in rules:
        subset &= np.asarray(r, dtype=bool)
    return subset, rr_z, rr_spectype, rr_subtype


# ------------------------------------------------------ resolution matrices
def resolution_mat_torows(mat):
    """desi_fit.py:682-685: [w, npix] DESI column band -> rows;
    out[r, j] = R[j, j + r - w//2] (np.roll wrap-around at the two ends)"""
    w = mat.shape[0]
    w2 = w // 2
    return np.array([np.roll(mat[_], _ - w2) for _ in range(w)])[::-1]


def resolution_mat_tocolumns(mat):
    """desi_fit.py:688-691: inverse of resolution_mat_torows"""
    w = mat.shape[0]
    w2 = w // 2
    return np.array([np.roll(mat[::-1][_], w2 - _) for _ in range(w)])


def _gau_mat(width, sigma0_angstrom, pix_size_angstrom):
    sig_pix = sigma0_angstrom / pix_size_angstrom
    xs = np.arange(width)
    return np.array([
        1. / np.sqrt(2 * np.pi) / sig_pix * np.exp(-0.5 *
                                                   ((xs - i) / sig_pix)**2)
        for i in range(len(xs))
    ])


def deconvolve_resolution_matrix(mat0, sigma0_angstrom=0.5,
                                 pix_size_angstrom=0.8):
    """desi_fit.py:694-720: take the template LSF (Gaussian sigma0) out of the
    DESI resolution band, row by row"""
    width, npix = mat0.shape
    gau_mat = _gau_mat(width, sigma0_angstrom, pix_size_angstrom)
    w2 = width // 2
    mat_rows = resolution_mat_torows(mat0)
    for i in range(w2):
        mat_rows[:w2 - i - 1, i] = 0
        j = npix - 1 - i
        mat_rows[w2 + 1 + i:, j] = 0
    mat_rows1 = scipy.linalg.solve(gau_mat, mat_rows)
    return resolution_mat_tocolumns(mat_rows1)


def _renormalise_rows(mat_rows):
    """the edge renormalisation of desi_fit.py:735-745 on [..., w, npix] rows"""
    w, npix = mat_rows.shape[-2:]
    w2 = w // 2
    mult = np.median(mat_rows.sum(axis=-2), axis=-1)
    mult = np.where(mult == 0, 1, mult)
    for i in range(w2):
        N1 = mat_rows[..., w2 - i:, i].sum(axis=-1)
        mat_rows[..., :, i] = mat_rows[..., :, i] / (
            N1 + (N1 == 0))[..., None] * mult[..., None]
        j = npix - 1 - i
        N2 = mat_rows[..., :w2 + 1 + i, j].sum(axis=-1)
        mat_rows[..., :, j] = mat_rows[..., :, j] / (
            N2 + (N2 == 0))[..., None] * mult[..., None]
    return mat_rows


def construct_resolution_sparse_matrix(mat, pix_size_angstrom=None,
                                       sigma0_angstrom=None):
    """desi_fit.py:723-748: scipy dia_matrix [npix, npix]"""
    width, npix = mat.shape
    w2 = width // 2
    mat = deconvolve_resolution_matrix(mat.copy(),
                                       pix_size_angstrom=pix_size_angstrom,
                                       sigma0_angstrom=sigma0_angstrom)
    mat_rows = _renormalise_rows(resolution_mat_torows(mat))
    mat = resolution_mat_tocolumns(mat_rows)
    return scipy.sparse.dia_matrix((mat, np.arange(w2, -w2 - 1, -1)),
                                   (npix, npix))


def resolution_row_taps(mats, pix_size_angstrom, sigma0_angstrom):
    """construct_resolution_sparse_matrix for a stack of fibres, straight into
    the row-tap layout of the kernels (engine.resol_taps):
    mats [n, w, npix] -> taps float64 [n, npix, w], taps[s, k, d] = R_s[k, k-w2+d]
    (entries that fall outside the matrix are zero, as in the dia matrix)."""
    mats = np.asarray(mats)
    n, w, npix = mats.shape
    w2 = w // 2
    # torows for the stack: rows[:, r, j] = mats[:, w-1-r, j + r - w2] (wrapped)
    rows = np.stack([np.roll(mats[:, w - 1 - r], (w - 1 - r) - w2, axis=-1)
                     for r in range(w)], axis=1)
    for i in range(w2):
        rows[:, :w2 - i - 1, i] = 0
        rows[:, w2 + 1 + i:, npix - 1 - i] = 0
    gau = _gau_mat(w, sigma0_angstrom, pix_size_angstrom)
    rows = np.linalg.solve(gau[None], rows.astype(np.float64))
    # the reference goes rows -> columns -> rows between the two steps; the
    # round trip is the identity (pure index shuffles)
    rows = _renormalise_rows(rows)
    taps = np.ascontiguousarray(np.swapaxes(rows, 1, 2))
    k = np.arange(npix)[:, None] + np.arange(-w2, w2 + 1)[None, :]
    taps[:, (k < 0) | (k >= npix)] = 0
    return taps


# ------------------------------------------------------------ conditioning
def interpolate_bad_regions(spec, mask):
    """desi_fit.py:751-778: linear interpolation over masked runs, constant
    continuation at the two ends (one spectrum)"""
    spec = np.asarray(spec)
    return _interp_bad_rows(spec[None, :], np.asarray(mask, dtype=bool)[None, :])[0]


def _interp_bad_rows(spec, mask):
    """interpolate_bad_regions on rows [n, npix]; dtype of `spec` is kept
    (np.interp works in float64 and the result is stored back)."""
    n, npix = spec.shape
    out = spec * 1
    good = ~mask
    rr, cc = np.nonzero(mask & good.any(axis=1)[:, None])
    if len(rr) == 0:
        return out
    # nearest good pixel to the left / right of every masked one, in its own row (-1 /
    # npix where there is none): a search of the masked positions in the sorted list of
    # good ones -- the masked pixels are a few per cent of the array
    gflat = np.flatnonzero(good)
    row0 = rr.astype(np.int64) * npix
    i = np.searchsorted(gflat, row0 + cc)
    p = gflat[np.maximum(i - 1, 0)]
    q = gflat[np.minimum(i, len(gflat) - 1)]
    L = np.where((i > 0) & (p >= row0), p - row0, -1)
    R = np.where((i < len(gflat)) & (q < row0 + npix), q - row0, npix)
    Lc, Rc = np.maximum(L, 0), np.minimum(R, npix - 1)
    yl = spec[rr, Lc].astype(np.float64)
    yr = spec[rr, Rc].astype(np.float64)
    with np.errstate(all='ignore'):
        slope = (yr - yl) / (Rc - Lc).astype(np.float64)
        mid = slope * (cc - Lc).astype(np.float64) + yl  # numpy's interp formula
    val = np.where(L < 0, yr, np.where(R >= npix, yl, mid))
    out[rr, cc] = val.astype(out.dtype)
    return out


LARGE_ERROR = 1000  # sets the error of masked pixels
MINERR_FRAC = 0.3  # errors below this times the median are clamped


def _medspec(spec, badmask):
    """median flux of one arm with the fall-backs of desi_fit.py:833-843;
    None when the arm has to be skipped"""
    if badmask.all():
        return None
    # first usable candidate of: all pixels, the good positive ones, |flux|
    candidates = (lambda: spec, lambda: spec[(spec > 0) & ~badmask],
                  lambda: np.abs(spec))
    med = 0.0
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')   # (empty selections)
        for i, pick in enumerate(candidates):
            med = np.nanmedian(pick())
            if i == 0 and med != 0:
                break            # the plain median is taken as it is ...
            if i > 0 and np.isfinite(med):
                break            # ... the fall-backs only when finite
    return med if (np.isfinite(med) and med != 0) else None


def _row_nanmedian(a):
    """np.nanmedian(a[i]) for every row, without the per-row Python call:
    sorted rows (NaNs last), the middle element or the mean of the two middle
    ones in the array's own dtype -- the numbers np.median produces."""
    n = a.shape[0]
    srt = np.sort(a, axis=1)
    cnt = (~np.isnan(a)).sum(axis=1)
    r = np.arange(n)
    lo = srt[r, np.maximum((cnt - 1) // 2, 0)]
    hi = srt[r, np.maximum(cnt // 2, 0)]
    with np.errstate(all='ignore'):
        med = (lo + hi) / a.dtype.type(2)
    med = np.where(lo == hi, lo, med)  # also keeps +-inf medians exact
    return np.where(cnt > 0, med, a.dtype.type(np.nan)), cnt


def get_specdata_batch(waves, fluxes, ivars, masks, resolutions, seqids, setups,
                       use_resolution_matrix=False, mask_dicroic=True,
                       lsf_sigma0_angstrom=None, as_float64=True):
    """get_specdata (desi_fit.py:781-888) for the fibres `seqids` at once.

    Returns {setup: dict(spec, espec float64 [n, npix]; badmask bool [n, npix];
    ok bool [n] -- False where the reference skips the arm (all masked, or an
    insane median); taps float64 [n, npix, w] or None)}.  `as_float64=False`
    leaves spec / espec in the file's own type (float32 in DESI coadds): the
    values the reference converts to float64 are exactly these, and the batch
    path converts on the device (half the bytes to select, join and upload)."""
    seqids = np.asarray(seqids, dtype=np.int64)
    out = {}
    for s in setups:
        spec = fluxes[s][seqids]       # (fancy indexing: already copies)
        curivars = ivars[s][seqids]
        badmask = masks[s][seqids] > 0
        n, npix = spec.shape
        med = np.ones(n, dtype=np.float64)
        rowmed, _ = _row_nanmedian(spec)
        ok = ~badmask.all(axis=1) & np.isfinite(rowmed) & (rowmed != 0)
        med[ok] = rowmed[ok]
        # median exactly zero: the reference's fall-backs, row by row (rare)
        for i in np.nonzero(~badmask.all(axis=1) & (rowmed == 0))[0]:
            m = _medspec(spec[i], badmask[i])
            if m is not None:
                med[i], ok[i] = m, True
        with np.errstate(all='ignore'):
            baddat = ~np.isfinite(spec + curivars)
        if mask_dicroic:
            dicroicmask = (waves[s] > 4300) & (waves[s] < 4450)
        else:
            dicroicmask = np.zeros(npix, dtype=bool)
        baderr = curivars <= 0
        edge_mask = np.zeros(npix, dtype=bool)
        taps = None
        if use_resolution_matrix:
            dwave = waves[s][1] - waves[s][0]
            taps = resolution_row_taps(resolutions[s][seqids], dwave,
                                       lsf_sigma0_angstrom[s])
            edge_pixels = 5  # the resolution matrix is corrupted at the edges
            edge_mask[:edge_pixels] = True
            edge_mask[-edge_pixels:] = True
        badall_interp = baddat | badmask | baderr
        badall = badall_interp | dicroicmask[None, :] | edge_mask[None, :]
        # 1. / medspec**2 / large_error**2 is float64 in the reference's numpy
        fill = (1. / med**2 / LARGE_ERROR**2).astype(curivars.dtype)
        curivars = np.where(badall, fill[:, None], curivars)
        spec = _interp_bad_rows(spec, badall_interp)
        with np.errstate(all='ignore'):
            espec = (1. / np.sqrt(curivars)).astype(curivars.dtype, copy=False)
        # sigma floor: 0.3 x the median error of the good pixels of the row
        ngood = (~badall).sum(axis=1)
        for i in np.nonzero(ok & (ngood == 0))[0]:
            logging.warning('The whole spectrum was masked...')
        gmed, _ = _row_nanmedian(np.where(badall, espec.dtype.type(np.nan),
                                          espec))
        thresh = gmed.astype(np.float64) * MINERR_FRAC
        # float32 array against a float64 scalar: compared in float32
        with np.errstate(all='ignore'):
            replace = (espec < thresh.astype(espec.dtype)[:, None]) & ~badall & \
                (ok & (ngood > 0))[:, None]
        nrep = replace.sum(axis=1)
        for i in np.nonzero(nrep / np.maximum(ngood, 1) > .01)[0]:
            logging.warning(
                'More than 1% of spectra had the uncertainty clamped')
        espec = np.where(replace, thresh.astype(espec.dtype)[:, None], espec)
        if as_float64:
            spec, espec = spec.astype(np.float64), espec.astype(np.float64)
        out[s] = dict(spec=spec, espec=espec, badmask=badall, ok=ok, taps=taps)
    return out


def get_specdata(waves, fluxes, ivars, masks, resolutions, seqid, setups,
                 use_resolution_matrix=False, mask_dicroic=True,
                 lsf_sigma0_angstrom=None):
    """desi_fit.py:781-888: tuple of spec_fit.SpecData for one fibre (arms that
    are fully masked or have an insane median are left out), or None."""
    c = get_specdata_batch(waves, fluxes, ivars, masks, resolutions, [seqid],
                           setups, use_resolution_matrix=use_resolution_matrix,
                           mask_dicroic=mask_dicroic,
                           lsf_sigma0_angstrom=lsf_sigma0_angstrom)
    sds = []
    for s in setups:
        a = c[s]
        if not a['ok'][0]:
            continue
        resol = None
        if use_resolution_matrix:
            resol = spec_fit.ResolMatrix(_taps_to_dia(a['taps'][0]))
        sds.append(spec_fit.SpecData('desi_%s' % s, waves[s], a['spec'][0],
                                     a['espec'][0], resolution=resol,
                                     badmask=a['badmask'][0]))
    if len(sds) == 0:
        logging.warning(f'No good data found for fiber {seqid}')
        return None
    return tuple(sds)


def _taps_to_dia(taps):
    """row taps [npix, w] -> scipy dia_matrix (R[k, k + o] = taps[k, w2 + o])"""
    npix, w = taps.shape
    w2 = w // 2
    data = np.zeros((w, npix))
    k = np.arange(npix)
    for di in range(w):
        o = w2 - di
        ok = (k + o >= 0) & (k + o < npix)
        data[di, k[ok] + o] = taps[k[ok], w2 + o]
    return scipy.sparse.dia_matrix((data, np.arange(w2, -w2 - 1, -1)),
                                   (npix, npix))


# ----------------------------------------------------------------- schema
def get_column_desc(setups):
    """desi_fit.py:910-959: {column: (dtype, comment)} -- the comments go to the
    TCOMMn cards of the RVTAB; the dtypes are informative (the reference writes
    the columns with the dtype of the values)."""
    columnDesc = dict([
        ('VRAD', (np.float32, 'Radial velocity')),
        ('VRAD_ERR', (np.float32, 'Radial velocity error')),
        ('VRAD_SKEW', (np.float32, 'Radial velocity posterior skewness')),
        ('VRAD_KURT', (np.float32, 'Radial velocity posterior kurtosis')),
        ('VSINI', (np.float32, 'Stellar rotation velocity')),
        ('LOGG', (np.float32, 'Log of surface gravity')),
        ('TEFF', (np.float32, 'Effective temperature')),
        ('FEH', (np.float32, '[Fe/H] from template fitting')),
        ('ALPHAFE', (np.float32, '[alpha/Fe] from template fitting')),
        ('LOGG_ERR', (np.float32, 'Log of surface gravity uncertainty')),
        ('TEFF_ERR', (np.float32, 'Effective temperature uncertainty')),
        ('FEH_ERR', (np.float32, '[Fe/H] uncertainty from template fitting')),
        ('ALPHAFE_ERR', (np.float32,
                         '[alpha/Fe] uncertainty from template fitting')),
        ('CHISQ_TOT', (np.float64, 'Total chi-square for all arms')),
        ('NPIX_TOT', (np.float64, 'Total number of unmasked pixels fitted')),
        ('CHISQ_C_TOT',
         (np.float64, 'Total chi-square for all arms for polynomial only fit')),
        ('CHISQ_CCF', (np.float32, 'Total chi-square from CCF fit')),
        ('TEFF_CCF', (np.float32, 'Effective temperature from CCF fit')),
        ('LOGG_CCF', (np.float32, 'Log of surface gravity from CCF fit')),
        ('FEH_CCF', (np.float32, '[Fe/H] from CCF fit')),
        ('ALPHAFE_CCF', (np.float32, '[alpha/Fe] from CCF fit')),
        ('VSINI_CCF', (np.float32, 'Vsini from CCF fit')),
        ('VRAD_CCF', (np.float32, 'Initial velocity from cross-correlation')),
        ('TARGETID', (np.int64, 'DESI targetid')),
        ('EXPID', (np.int64, 'DESI exposure id')),
        ('SUCCESS', (bool, 'Did we succeed or fail')),
        ('RVS_WARN', (np.int64, 'RVSpecFit warning flag')),
        ('RR_Z', (np.float64, 'Redrock redshift')),
        ('RR_SPECTYPE', (str, 'Redrock spectype')),
        ('RR_SUBTYPE', (str, 'Redrock spectroscopic subtype'))
    ])
    for curs in setups:
        curs = curs.upper()
        columnDesc['SN_%s' % curs] = (np.float32,
                                      'Median S/N in the %s arm' % curs)
        columnDesc['CHISQ_%s' % curs] = (np.float64,
                                         'Chi-square in the %s arm' % curs)
        columnDesc['CHISQ_C_%s' % curs] = (
            np.float64,
            'Chi-square in the %s arm after fitting continuum only' % curs)
    return columnDesc


def comment_filler(tab, desc):
    """desi_fit.py:891-900"""
    for i, name in enumerate(tab.data.columns.names):
        comm = desc.get(name)
        tab.header['TCOMM%d' % (i + 1)] = '' if comm is None else comm[1]
    return tab


def put_empty_file(fname):
    """desi_fit.py:903-907"""
    pyfits.HDUList([pyfits.PrimaryHDU(header=get_prim_header())]).writeto(
        fname, overwrite=True, checksum=True)


def write_hdulist(fname, hdulist):
    """desi_fit.py:1302-1308: write through a temporary file"""
    fname_tmp = fname + '.tmp'
    hdulist.writeto(fname_tmp, overwrite=True, checksum=True)
    os.rename(fname_tmp, fname)


def rows_to_table(rows):
    """list of dicts -> fits_min.FitsTable, the way astropy.table.Table(rows)
    + BinTableHDU lay it out (desi_fit.py:1262, 1288): columns in order of
    first appearance, cells a row does not have are NaN (floats), 999999
    (integers, TNULL), '' (strings), False (logicals)."""
    names = []
    for r in rows:
        for k in r:
            if k not in names:
                names.append(k)
    tab = pyfits.FitsTable()
    for k in names:
        vals = [r[k] for r in rows if k in r]
        proto = np.asarray(vals)
        kind = proto.dtype.kind
        if kind in 'SU':
            col = np.array([str(r.get(k, '')) for r in rows])
            if col.dtype.itemsize == 0:
                col = col.astype('U1')
            null = None
        elif kind == 'b':
            col = np.array([bool(r.get(k, False)) for r in rows])
            null = None
        elif kind in 'iu':
            col = np.array([r.get(k, INT_NULL) for r in rows], dtype=proto.dtype)
            null = INT_NULL
        else:
            col = np.array([r.get(k, np.nan) for r in rows], dtype=proto.dtype)
            null = None
        tab._cols.append(pyfits.Column(k, col, COLUMN_UNITS.get(k, ''),
                                       null=null))
    return tab


# -------------------------------------------------------------- the fits
NAME_MAPPINGS = (('logg', 'LOGG'), ('teff', 'TEFF'), ('feh', 'FEH'),
                 ('alpha', 'ALPHAFE'))


# RVS_DESI_STAGE_TIMES=1: wall seconds of fit_batch's stages (device synchronised at
# each boundary -- a measuring aid, tools/perf/desi_fpb.sh)
FIT_TIMES = {}


def fit_batch(batch, config, options, ccf_init=True):
    """The body of proc_onespec (desi_fit.py:283-354) for a SpecBatch: the
    starting point (CCF or brute-force grid), vel_fit.process, the continuum
    chi^2.  Returns host arrays (dict) with a leading S axis plus `yfit`
    (list over arms of [S, npix] float64)."""
    import torch
    from .. import fitter_ccf, vel_fit, _lib
    S = batch.S
    timed = bool(os.environ.get('RVS_DESI_STAGE_TIMES'))
    t_last = [time.time()]

    def tick(k):
        if timed:
            torch.cuda.synchronize()
            now = time.time()
            FIT_TIMES[k] = FIT_TIMES.get(k, 0.) + now - t_last[0]
            t_last[0] = now
    names = spec_inter.getSpecParams(batch.names[0], config)
    if ccf_init:
        res = fitter_ccf.fit(batch, config)
        if bool((res['status'] & _lib.ST_CCF_FAILED).any().item()):
            logging.error('Cross-correlation failed')
            raise RuntimeError('Cross-correlation step failed')
        pd0 = {k: res['best_par'][:, i].contiguous()
               for i, k in enumerate(names)}
        vs0 = res['best_vsini']
        vrad_ccf = res['best_vel'].cpu().numpy()
    else:
        g = vel_fit.firstguess(batch, config=config, options=options)
        pd0 = {k: g[k].contiguous() for k in names}
        vs0 = g['vsini']
        vrad_ccf = None
    tick('start_point')
    has_vs = torch.isfinite(vs0)
    groups = []
    if bool(has_vs.all().item()):
        groups.append((None, True))
    elif not bool(has_vs.any().item()):
        groups.append((None, False))
    else:  # templates with and without rotation: two lock-step runs
        groups.append((torch.nonzero(has_vs).reshape(-1), True))
        groups.append((torch.nonzero(~has_vs).reshape(-1), False))
    out = dict(
        vel=np.full(S, np.nan), vel_err=np.full(S, np.nan),
        vel_skewness=np.full(S, np.nan), vel_kurtosis=np.full(S, np.nan),
        vsini=np.full(S, np.nan), bad_hessian=np.zeros(S, dtype=bool),
        chisq_array=np.zeros((S, len(batch.arms))),
        npix_array=np.zeros((S, len(batch.arms)), dtype=np.int64),
        param={k: np.full(S, np.nan) for k in names},
        param_err={k: np.full(S, np.nan) for k in names})
    yfit = [np.zeros((S, a.npix)) for a in batch.arms]
    for idx, with_vs in groups:
        sub = batch if idx is None else batch.subset(idx)
        sel = slice(None) if idx is None else idx.cpu().numpy()
        p0 = {k: (v if idx is None else v[idx].contiguous())
              for k, v in pd0.items()}
        if with_vs:
            p0['vsini'] = (vs0 if idx is None else vs0[idx]).contiguous()
        r = vel_fit.process(sub, p0, fixParam=[], config=config,
                            options=options)
        tick('process')
        for k in ('vel', 'vel_err', 'vel_skewness', 'vel_kurtosis'):
            out[k][sel] = r[k].cpu().numpy()
        if 'vsini' in r and r['vsini'] is not None:
            out['vsini'][sel] = r['vsini'].cpu().numpy()
        out['bad_hessian'][sel] = np.asarray(r['bad_hessian'])
        out['chisq_array'][sel] = r['chisq_array'].cpu().numpy()
        out['npix_array'][sel] = r['npix_array'].cpu().numpy()
        for k in names:
            out['param'][k][sel] = r['param'][k].cpu().numpy()
            out['param_err'][k][sel] = np.asarray(r['param_err'][k])
        for ia in range(len(batch.arms)):
            yfit[ia][sel] = r['yfit'][ia].cpu().numpy()
        tick('to_host')
    cont = spec_fit.get_chisq_continuum(batch, options=options)['chisq_array']
    out['chisq_c_array'] = cont.cpu().numpy()
    tick('continuum')
    out['vrad_ccf'] = vrad_ccf
    out['yfit'] = yfit
    return out


def _outdicts(fr, arm_names, config, ccf_init):
    """outdict of proc_onespec (desi_fit.py:312-356) for every spectrum of a
    fit_batch result; arm_names are the SpecData names ('desi_b', ...)."""
    S = len(fr['vel'])
    rows = []
    tags = [n.replace('desi_', '').upper() for n in arm_names]
    chisq_tot = fr['chisq_array'].sum(axis=1)
    chisq_c_tot = fr['chisq_c_array'].sum(axis=1)
    p = fr['param']
    warn = rvs_warn_bits(chisq_tot, chisq_c_tot, fr['vel'], fr['vsini'],
                         fr['vel_err'], fr['bad_hessian'],
                         p.get('teff', np.full(S, 5000.)),
                         p.get('feh', np.zeros(S)), p.get('logg', np.full(S, 3.)),
                         config)
    for i in range(S):
        d = dict(VRAD=fr['vel'][i], VRAD_ERR=fr['vel_err'][i],
                 VRAD_SKEW=fr['vel_skewness'][i],
                 VRAD_KURT=fr['vel_kurtosis'][i], VSINI=fr['vsini'][i])
        for name1, name2 in NAME_MAPPINGS:
            if name1 in p:
                d[name2] = p[name1][i]
                d[name2 + '_ERR'] = fr['param_err'][name1][i]
        d['CHISQ_TOT'] = chisq_tot[i]
        d['CHISQ_C_TOT'] = chisq_c_tot[i]
        d['NPIX_TOT'] = int(fr['npix_array'][i].sum())
        for ia, t in enumerate(tags):
            d['CHISQ_%s' % t] = fr['chisq_array'][i, ia]
            d['CHISQ_C_%s' % t] = float(fr['chisq_c_array'][i, ia])
        if ccf_init:
            d['VRAD_CCF'] = fr['vrad_ccf'][i]
        d['RVS_WARN'] = int(warn[i])
        rows.append(d)
    return rows


def _template_versions(names, config):
    """desi_fit.py:372-376: every interpolator in the process-wide cache (not
    only the arms of this fibre), in the order they were loaded"""
    for n in names:
        spec_inter.getInterpolator(n, config)
    return {
        k: dict(revision=v.revision,
                creation_soft_version=v.creation_soft_version)
        for k, v in spec_inter.interp_cache.interps.items()
    }


def proc_onespec(specdata, setups, config, options, resolution_matrix=None,
                 fig_fname='fig.png', ccf_init=True, doplot=True):
    """desi_fit.py:248-378 for one fibre (a tuple of SpecData): returns
    (outdict, yfit).  Values are plain floats; the units the reference attaches
    are in COLUMN_UNITS."""
    from ..spec_fit import as_batch
    batch, _ = as_batch(list(specdata))
    fr = fit_batch(batch, config, options, ccf_init=ccf_init)
    outdict = _outdicts(fr, batch.names, config, ccf_init)[0]
    if doplot:
        logging.warning('plots are not produced by this build')
    outdict['versions'] = _template_versions(batch.names, config)
    return outdict, [y[0] for y in fr['yfit']]


def _arm_batch(cond, setups, pattern, rows, waves, device):
    """SpecBatch of the fibres `rows` (indices into the conditioned arrays)
    over the arms of `pattern`"""
    from .. import engine
    arms = []
    for s, use in zip(setups, pattern):
        if not use:
            continue
        c = cond[s]
        a = engine.ArmData('desi_%s' % s, waves[s], c['spec'][rows],
                           c['espec'][rows],
                           c['badmask'][rows].astype(np.uint8), device=device)
        if c['taps'] is not None:
            t = c['taps'][rows]
            a.resol = engine.make_resol(t, t.shape[2], len(rows), device)
        arms.append(a)
    return engine.SpecBatch(arms)


def _proc_desi_steps(fname, tab_ofname, mod_ofname, fig_prefix, config,
                     fit_targetid=None, objtypes=None, doplot=True, minsn=-1e9,
                     expid_range=None, poolex=None, fitarm=None, cmdline=None,
                     zbest_select=False, zbest_include=False,
                     use_resolution_matrix=False, ccf_init=True, npoly=10,
                     device='cuda', max_batch=4096, timers=None):
    """proc_desi as a generator: everything up to the conditioned spectra, then
    ONE yield of a fit request (see _fit_requests) that is answered with the
    per-fibre results, then the products.  The return value (StopIteration
    .value) is proc_desi's.  Files that need no fit return before the yield."""
    if npoly is None:
        npoly = 10
    tm = timers if timers is not None else {}
    t_last = [time.time()]

    def tick(k):
        now = time.time()
        tm[k] = tm.get(k, 0.) + now - t_last[0]
        t_last[0] = now
    logging.info('Processing %s', fname)
    FP = None
    try:
        FP = pyfits.open(fname)
    except OSError:
        pass
    if FP is None or not valid_file(FP):
        logging.error('%s file: %s', 'Cannot read' if FP is None else 'Not valid',
                      fname)
        return -1
    setups = [a for a in ('b', 'r', 'z') if fitarm is None or a in fitarm]
    assert setups, 'fitarm selects no arm'
    spectrum_header = FP[0].header
    fibermap, scores = FP['FIBERMAP'].data, FP['SCORES'].data
    exp_fibermap = FP['EXP_FIBERMAP'].data if 'EXP_FIBERMAP' in FP else None
    if fit_targetid is not None and \
            not np.isin(fibermap['TARGETID'], fit_targetid).any():
        # nothing asked for lives in this file: empty products mark it as done
        logging.warning('No fibers selected in file %s', fname)
        for ofname in (tab_ofname, mod_ofname):
            put_empty_file(ofname)
        return 0
    fluxes, ivars, masks, waves, resolutions = read_data(FP, setups)
    tick('read')
    sns = _arm_sns(scores, setups, fluxes, ivars, masks)
    short = [a for a in setups if len(sns[a]) != len(fibermap)]
    if short:
        logging.warning('file %s: arm(s) %s hold another number of spectra than '
                        'the fibermap has rows; skipping the file', fname,
                        ','.join(short))
        return -1
    columnDesc = get_column_desc(setups)
    zbest_path, zbest_ext = get_zbest_fname(fname) \
        if (zbest_select or zbest_include) else (None, None)
    subset, rr_z, rr_spectype, rr_subtype = select_fibers_to_fit(
        fibermap, sns, minsn=minsn, objtypes=objtypes, expid_range=expid_range,
        fit_targetid=fit_targetid, zbest_path=zbest_path, zbest_ext=zbest_ext,
        zbest_select=zbest_select, zbest_include=zbest_include)

    fibermap_subset_hdu = pyfits.BinTableHDU(fibermap[subset], name='FIBERMAP')
    exp_fibermap_subset_hdu = None
    if exp_fibermap is not None:
        tmp_sub = np.isin(exp_fibermap['TARGETID'],
                          fibermap['TARGETID'][subset])
        exp_fibermap_subset_hdu = pyfits.BinTableHDU(exp_fibermap[tmp_sub],
                                                     name='EXP_FIBERMAP')
    scores_subset_hdu = pyfits.BinTableHDU(scores[subset], name='SCORES')
    tick('select')

    def mod_hdus(versions, models):
        hdus = [pyfits.PrimaryHDU(header=get_prim_header(
            versions=versions, config=config, cmdline=cmdline,
            spectrum_header=spectrum_header, zbest_path=zbest_path))]
        for curs in setups:
            hdus.append(pyfits.ImageHDU(waves[curs],
                                        name='%s_WAVELENGTH' % curs.upper()))
            hdus.append(pyfits.ImageHDU(
                None if models is None else models['desi_%s' % curs],
                name='%s_MODEL' % curs.upper()))
        return hdus + [fibermap_subset_hdu]

    def tab_hdus(versions, outtab):
        hdus = [pyfits.PrimaryHDU(header=get_prim_header(
            versions=versions, config=config, cmdline=cmdline,
            zbest_path=zbest_path)),
            comment_filler(pyfits.BinTableHDU(outtab, name='RVTAB'),
                           columnDesc), fibermap_subset_hdu, scores_subset_hdu]
        if exp_fibermap_subset_hdu is not None:
            hdus.append(exp_fibermap_subset_hdu)
        return hdus

    if not subset.any():
        logging.warning('No fibers selected in file %s' % (fname))
        write_hdulist(mod_ofname, pyfits.HDUList(mod_hdus(None, None)))
        write_hdulist(tab_ofname,
                      pyfits.HDUList(tab_hdus(None, pyfits.FitsTable())))
        return 0
    logging.info('Selected %d fibers to fit' % (subset.sum()))

    columnsCopy = ['FIBER', 'REF_ID', 'REF_CAT', 'TARGET_RA', 'TARGET_DEC',
                   'TARGETID', 'EXPID']
    seqid_to_fit = np.nonzero(subset)[0]
    nsel = len(seqid_to_fit)
    if rr_z is not None:
        rr_z, rr_spectype, rr_subtype = (rr_z[seqid_to_fit],
                                         rr_spectype[seqid_to_fit],
                                         rr_subtype[seqid_to_fit])
    else:
        rr_z = np.zeros(nsel) + np.nan
        rr_spectype = np.zeros(nsel, dtype='U1')
        rr_subtype = np.zeros(nsel, dtype='U1')
    sig0s = None
    if use_resolution_matrix:
        sig0s = {}
        for s in setups:
            if ('lsf_sigma0_angstrom' not in config
                    or s not in config['lsf_sigma0_angstrom']):
                sig0s[s] = 0.5
                logging.warning('sigma0 of the templates is not specified '
                                f'for setup {s} using {sig0s[s]}')
            else:
                sig0s[s] = config['lsf_sigma0_angstrom'][s]
    if doplot:
        logging.warning('plots are not produced by this build')

    # ---- conditioning of all selected fibres, then one batch per arm pattern
    cond = get_specdata_batch(waves, fluxes, ivars, masks, resolutions,
                              seqid_to_fit, setups,
                              use_resolution_matrix=use_resolution_matrix,
                              lsf_sigma0_angstrom=sig0s, as_float64=False)
    okmat = np.stack([cond[s]['ok'] for s in setups], axis=1)
    tick('condition')
    for i in np.nonzero(~okmat.any(axis=1))[0]:
        logging.warning('No good data found for fiber %d' % seqid_to_fit[i])
    outdicts, curmodels, arms_of = yield dict(
        cond=cond, okmat=okmat, setups=setups, waves=waves, nsel=nsel)
    tick('fit')
    nfibers_good = sum(_ is not None for _ in outdicts)
    good_flags = [_ is not None for _ in outdicts]
    # the reference indexes `models` (nfibers_good rows) by the ROW counter
    # (desi_fit.py:1211-1252): identical shapes whenever it does not raise
    # (no BAD_SPECTRUM fibre ahead of a good one); nsel rows otherwise
    last_good = max([i for i, g in enumerate(good_flags) if g], default=-1)
    nmod = nfibers_good if last_good < nfibers_good else nsel
    models = {'desi_' + s: np.zeros((nmod, len(waves[s])), dtype=np.float32)
              for s in setups}
    versions = None
    outdf = []
    for ii in range(nsel):
        outdict = outdicts[ii]
        bad_row = outdict is None
        if bad_row:
            outdict = dict(RVS_WARN=bitmasks['BAD_SPECTRUM'])
        cur_seqid = seqid_to_fit[ii]
        for col in columnsCopy:
            if col in fibermap.columns.names:
                outdict[col] = fibermap[col][cur_seqid]
        for curs in setups:
            outdict['SN_%s' % curs.upper()] = sns[curs][cur_seqid]
        outdict['SUCCESS'] = outdict['RVS_WARN'] == 0
        outdict['RR_Z'] = rr_z[ii]
        outdict['RR_SPECTYPE'] = rr_spectype[ii]
        outdict['RR_SUBTYPE'] = rr_subtype[ii]
        if not bad_row:
            for jj, curs in enumerate(arms_of[ii]):
                models[curs][ii] = curmodels[ii][jj]
            if versions is None:
                versions = _template_versions(arms_of[ii], config)
        outdf.append(outdict)
    outtab = rows_to_table(outdf)
    assert (len(fibermap_subset_hdu.data) == len(outtab))
    write_hdulist(mod_ofname, pyfits.HDUList(mod_hdus(versions, models)))
    write_hdulist(tab_ofname, pyfits.HDUList(tab_hdus(versions, outtab)))
    tick('write')
    return nsel


def _stage_requests(reqs, device='cuda', max_batch=4096):
    """The host half of _fit_requests: files that share arms and wavelength grids
    concatenated, one SpecBatch per pattern of usable arms (in chunks of
    `max_batch`), uploaded.  Needs nothing from the GPU, so proc_many's worker
    thread does it for group g + 1 while group g is fitted (on a stream of its
    own: the copies do not queue behind the fit's kernels).  Returns
    [(batch, rows, names_p, owner, local)]."""
    staged = []
    groups = {}
    for ir, r in enumerate(reqs):
        key = (tuple(r['setups']),
               tuple(r['waves'][s].tobytes() for s in r['setups']),
               tuple(r['cond'][s]['taps'] is not None for s in r['setups']))
        groups.setdefault(key, []).append(ir)
    for (setups, _, _), members in groups.items():
        setups = list(setups)
        waves = reqs[members[0]]['waves']
        okmat = np.concatenate([reqs[ir]['okmat'] for ir in members], axis=0)
        owner = np.concatenate([np.full(reqs[ir]['nsel'], ir) for ir in members])
        local = np.concatenate([np.arange(reqs[ir]['nsel']) for ir in members])
        if len(members) == 1:
            cond = reqs[members[0]]['cond']
        else:
            cond = {}
            for s in setups:
                cs = [reqs[ir]['cond'][s] for ir in members]
                cond[s] = {k: (None if cs[0][k] is None else
                               np.concatenate([c[k] for c in cs], axis=0))
                           for k in ('spec', 'espec', 'badmask', 'ok', 'taps')}
        for pattern in sorted({tuple(_) for _ in okmat.tolist()}, reverse=True):
            if not any(pattern):
                continue
            rows_p = np.nonzero(
                (okmat == np.array(pattern)[None, :]).all(axis=1))[0]
            names_p = ['desi_%s' % s for s, u in zip(setups, pattern) if u]
            for c0 in range(0, len(rows_p), max_batch):
                rows = rows_p[c0:c0 + max_batch]
                staged.append((_arm_batch(cond, setups, pattern, rows, waves, device),
                               rows, names_p, owner, local))
    return staged


def _fit_requests(reqs, config, options, ccf_init=True, device='cuda',
                  max_batch=4096, staged=None):
    """Fit the conditioned fibres of one or several files (the requests that
    _proc_desi_steps yields).  Files that share arms and wavelength grids are
    fitted TOGETHER -- one batch per pattern of usable arms -- because the
    lock-step optimiser is bound by round latency at a few hundred fibres
    (a 500-fibre coadd) and by throughput from a few thousand on.  Every fibre's
    result is independent of what else is in its batch.  `staged`: the batches,
    if _stage_requests has already built them.
    Returns, per request, (outdicts, curmodels, arms_of): lists over its fibres
    (None for fibres without a usable arm)."""
    out = [([None] * r['nsel'], [None] * r['nsel'], [None] * r['nsel'])
           for r in reqs]
    if staged is None:
        staged = _stage_requests(reqs, device=device, max_batch=max_batch)
    for batch, rows, names_p, owner, local in staged:
        fr = fit_batch(batch, config, options, ccf_init=ccf_init)
        for k, (i, d) in enumerate(zip(rows, _outdicts(
                fr, names_p, config, ccf_init))):
            o = out[owner[i]]
            o[0][local[i]] = d
            o[1][local[i]] = [y[k] for y in fr['yfit']]
            o[2][local[i]] = names_p
    return out


def proc_desi(fname, tab_ofname, mod_ofname, fig_prefix, config, **kwargs):
    """desi_fit.py:962-1299: fit every selected fibre of one DESI file and write
    the RVTAB and RVMOD products (same keyword arguments as the reference, plus
    device / max_batch / timers).  `poolex` is accepted for signature
    compatibility and unused: the fibres of the file are one GPU batch (in
    chunks of `max_batch`).  Returns the number of fibres selected, or -1.
    `timers` (dict) receives the wall seconds of the stages read / select /
    condition / fit / write."""
    return proc_desi_group([(fname, tab_ofname, mod_ofname, fig_prefix)], config,
                           **kwargs)[0]


def _group_prepare(files, config, kwargs):
    """read, select and condition every file of the group (host only): the
    generators parked at their fit request, and the return values of the files
    that needed no fit"""
    gens, rets = [], [None] * len(files)
    kw = {k: v for k, v in kwargs.items() if k != 'stage_ahead'}
    for i, f in enumerate(files):
        g = _proc_desi_steps(*f, config, **kw)
        try:
            gens.append((i, g, next(g)))
        except StopIteration as e:
            rets[i] = e.value
    if kwargs.get('stage_ahead') and gens:
        # (proc_many's worker thread: the group's batches built and uploaded while
        # the group before is fitted)
        import torch
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            staged = _stage_requests([r for _, _, r in gens],
                                     device=kwargs.get('device', 'cuda'),
                                     max_batch=kwargs.get('max_batch', 4096))
        side.synchronize()
        return gens, rets, staged
    return gens, rets


def _group_fit(state, config, kwargs):
    """fit the parked requests of a prepared group together (the GPU part)"""
    gens = state[0]
    npoly = kwargs.get('npoly')
    options = {'npoly': 10 if npoly is None else npoly}
    return _fit_requests([r for _, _, r in gens], config, options,
                         ccf_init=kwargs.get('ccf_init', True),
                         device=kwargs.get('device', 'cuda'),
                         max_batch=kwargs.get('max_batch', 4096),
                         staged=state[2] if len(state) > 2 else None)


def _group_write(state, res, errors=None):
    """hand every file of a fitted group its results: the generators assemble and
    write the products (host work only).  With `errors` (a dict) a file whose
    generator raises does not stop the others: its exception is recorded under
    its index and its return value stays None"""
    gens, rets = state[0], state[1]
    for (i, g, _), r in zip(gens, res):
        try:
            g.send(r)
            raise RuntimeError('proc_desi generator did not finish')
        except StopIteration as e:
            rets[i] = e.value
        except Exception as e:  # noqa: BLE001 -- reported per file
            if errors is None:
                raise
            errors[i] = e
    return rets


def _group_finish(state, config, kwargs):
    """fit the parked requests together and let every file write its products"""
    return _group_write(state, _group_fit(state, config, kwargs))


def proc_desi_group(files, config, **kwargs):
    """proc_desi for several files at once: `files` is a list of (fname,
    tab_ofname, mod_ofname, fig_prefix); the selected fibres of ALL of them go
    through the GPU together (_fit_requests), the products are written per file
    as by proc_desi.  Returns the list of proc_desi return values."""
    return _group_finish(_group_prepare(files, config, kwargs), config, kwargs)


def proc_desi_wrapper(*args, **kwargs):
    """proc_desi for one file with its outcome recorded: a line of the status
    file (SUCCESS with the number of fitted fibres, FAILURE otherwise; the
    reference's desi_fit.py:1311-1345 keeps the same record) and, unless
    `throw_exceptions`, an exception is logged and the file loop goes on.  The
    reference's per-process crash_<pid>.log files belong to its worker-pool
    control plane and are not written: the traceback goes to the log."""
    status_file = kwargs.pop('process_status_file', None)
    throw_exceptions = kwargs.pop('throw_exceptions', None)
    t0 = time.time()
    nfit, failed = 0, True
    try:
        ret = proc_desi(*args, **kwargs)
        failed = ret is not None and ret < 0
        nfit = 0 if (ret is None or failed) else ret
    except Exception:  # noqa: BLE001
        logging.exception('proc_desi failed on %s', args[0] if args else '?')
        if throw_exceptions:
            raise
    finally:
        if status_file is not None:
            update_process_status_file(
                status_file, args[0],
                ProcessStatus.FAILURE if failed else ProcessStatus.SUCCESS, nfit,
                time.time() - t0)


def _select_rank_device(environ=None):
    """One process per GPU: under torch.distributed.run every rank -- and every
    worker process a rank spawns, which inherits its environment -- fits on the
    GPU LOCAL_RANK names, whatever file shard it was given.  Returns the index
    selected (None: no launcher, the current device stays)."""
    environ = os.environ if environ is None else environ
    lr = environ.get('LOCAL_RANK')
    if lr is None:
        return None
    import torch
    idx = int(lr)
    n = torch.cuda.device_count()
    if n > 0 and idx >= n:
        # more ranks than visible GPUs: sharing a device silently would hide a
        # launcher mistake; same opt-in as bench.py
        if not environ.get('RVS_SHARE_GPU'):
            raise RuntimeError(
                'LOCAL_RANK=%d but only %d GPU(s) are visible; set '
                'RVS_SHARE_GPU=1 to let ranks share a device' % (idx, n))
        logging.warning('LOCAL_RANK=%d shares GPU %d (RVS_SHARE_GPU)', idx,
                        idx % n)
        idx %= n
    torch.cuda.set_device(idx)
    return idx


# seconds of the last proc_many call's file groups, by where the calling thread (or the
# writer thread: 'write') spent them -- what bench.py reports beside the fibre rate
GROUP_TIMES = dict(wait_prepare=0., fit=0., write=0., drain=0., groups=0, flushed=0)


def proc_many(files, output_dir, output_tab_prefix, output_mod_prefix,
              figure_dir=None, figure_prefix=None, config_fname=None,
              nthreads=1, fit_targetid=None, objtypes=None, minsn=-1e9,
              doplot=True, expid_range=None, skipexisting=False, fitarm=None,
              cmdline=None, zbest_select=False, zbest_include=False,
              ccf_init=True, subdirs=True, ccf_continuum_normalize=True,
              process_status_file=None, use_resolution_matrix=None, npoly=None,
              throw_exceptions=None, log_level=None, log_filename=None,
              shard=None, files_per_batch=8):
    """desi_fit.py:1392-1551: loop over files.  `nthreads` > 1 starts that many
    worker PROCESSES on this rank's GPU, each with its own stride of the rank's
    files (the reference's process pool, desi_fit.py:1475-1479): while one
    worker is in the latency-bound tail of its optimiser rounds the other's
    kernels fill the GPU -- two workers: +29 % fibres/s.  The template library
    must then be on disk (config['template_lib']), not only registered in
    memory.  `files_per_batch` files are read and conditioned one after the
    other and FITTED TOGETHER (proc_desi_group: a single 500-fibre coadd leaves
    the lock-step optimiser latency-bound); should the group fail, its files
    are retried one by one so that the failure lands on the file that caused
    it.  The first group is half a batch (from 4 files per batch up): its
    preparation is the one stretch of host work that no fit runs beside.  (Eight
    files -- 4000 fibres -- per batch since round 6: 16 files 1877 -> 1940 fibres/s
    against four, sixteen 1812; tools/perf/desi_fpb.sh.)  `shard=(rank, world)` -- or the RANK/WORLD_SIZE environment of
    torch.distributed.run -- gives every GPU process its own stride of the file
    list; files are independent, there is no collective."""
    override = dict(ccf_continuum_normalize=ccf_continuum_normalize)
    config = utils.read_config(config_fname, override)
    assert (config is not None)
    assert ('template_lib' in config)
    if shard is None:
        shard = (int(os.environ.get('RANK', 0)),
                 int(os.environ.get('WORLD_SIZE', 1)))
    rank, world = shard
    workers = nthreads is not None and nthreads > 1
    if not workers:
        # the process that fits selects its GPU; with worker processes the
        # parent only waits for them and never touches a GPU, and every worker
        # (shard (0, 1), environment inherited) comes through here itself
        _select_rank_device()
    if process_status_file is not None:
        if world > 1:
            process_status_file = '%s.%d' % (process_status_file, rank)
        update_process_status_file(process_status_file, None, None, None, None,
                                   start=True)
    if workers:
        import multiprocessing
        shared = getattr(files, 'shared', False)
        mine = None if shared else list(files)[rank::world]
        ctx = multiprocessing.get_context('spawn')
        procs = []
        for w in range(nthreads):
            # (a shared queue file: every worker takes its files from it)
            sub = files if shared else mine[w::nthreads]
            if not shared and not sub:
                continue
            kwa = dict(
                figure_dir=figure_dir, figure_prefix=figure_prefix,
                config_fname=config_fname, nthreads=1,
                fit_targetid=fit_targetid, objtypes=objtypes, minsn=minsn,
                doplot=doplot, expid_range=expid_range,
                skipexisting=skipexisting, fitarm=fitarm, cmdline=cmdline,
                zbest_select=zbest_select, zbest_include=zbest_include,
                ccf_init=ccf_init, subdirs=subdirs,
                ccf_continuum_normalize=ccf_continuum_normalize,
                process_status_file=None if process_status_file is None
                else '%s.w%d' % (process_status_file, w),
                use_resolution_matrix=use_resolution_matrix, npoly=npoly,
                throw_exceptions=throw_exceptions, log_level=log_level,
                log_filename=log_filename, shard=(0, 1),
                files_per_batch=files_per_batch)
            p = ctx.Process(target=proc_many, args=(
                sub, output_dir, output_tab_prefix, output_mod_prefix),
                kwargs=kwa)
            p.start()
            procs.append((w, p))
        bad = []
        for w, p in procs:
            p.join()
            if p.exitcode != 0:
                bad.append((w, p.exitcode))
            if process_status_file is not None:
                part = '%s.w%d' % (process_status_file, w)
                if os.path.exists(part):
                    with open(part) as fi, open(process_status_file, 'a') as fo:
                        fo.write(fi.read())
                    os.unlink(part)
        if bad:
            raise RuntimeError('worker processes failed: %s' % bad)
        logging.info('Successfully finished processing')
        return
    kw = dict(fit_targetid=fit_targetid, objtypes=objtypes, doplot=doplot,
              minsn=minsn, expid_range=expid_range, fitarm=fitarm,
              cmdline=cmdline, zbest_select=zbest_select,
              zbest_include=zbest_include, npoly=npoly, ccf_init=ccf_init,
              use_resolution_matrix=bool(use_resolution_matrix))
    # Groups of files_per_batch files.  While the GPU fits group g, a worker
    # thread reads and conditions group g + 1 (numpy and file I/O release the
    # GIL; the fit thread spends its time inside the library).
    import concurrent.futures
    import torch
    # (the worker thread also builds and uploads the group's batches -- where there
    # is a device to upload to)
    kw_prepare = dict(kw, stage_ahead=bool(torch.cuda.is_available()))
    GROUP_TIMES.update(wait_prepare=0., fit=0., write=0., drain=0., groups=0, flushed=0)
    pool = concurrent.futures.ThreadPoolExecutor(1)
    pending = []
    inflight = []   # [(group, future of _group_prepare)]

    def one_by_one(group):
        for f, t, m in group:
            proc_desi_wrapper(f, t, m, None, config,
                              process_status_file=process_status_file,
                              throw_exceptions=throw_exceptions, **kw)

    wpool = concurrent.futures.ThreadPoolExecutor(1)
    writing = []    # [(group, start time, future of _group_write)]

    def timed_write(state, res):
        """_group_write on the writer thread: (return values, {file index:
        exception}, seconds spent writing -- not waiting in the queue)"""
        t0 = time.time()
        errors = {}
        rets = _group_write(state, res, errors)
        GROUP_TIMES['write'] += time.time() - t0
        return rets, errors, time.time() - t0

    def collect_written(block=False):
        """groups whose products are written: status lines.  Only the files whose
        assembly / write raised are redone (file by file); the time recorded per
        file is the group's fit plus its write, not its wait for the writer"""
        while writing and (block or writing[0][2].done()):
            group, t_fit, wf = writing.pop(0)
            try:
                rets, errors, t_write = wf.result()
            except Exception:  # noqa: BLE001 -- retried per file
                logging.exception('writing a group of %d files failed; retrying '
                                  'one by one' % len(group))
                one_by_one(group)
                continue
            for i in sorted(errors):
                logging.error('writing the products of %s failed (%r); redoing '
                              'that file', group[i][0], errors[i])
            if process_status_file is not None:
                dt = (t_fit + t_write) / len(group)
                for i, ((f, _, _), n) in enumerate(zip(group, rets)):
                    if i in errors:
                        continue
                    update_process_status_file(
                        process_status_file, f,
                        ProcessStatus.SUCCESS if n is not None and n >= 0
                        else ProcessStatus.FAILURE, max(n or 0, 0), dt)
            if errors:
                one_by_one([group[i] for i in sorted(errors)])

    # RVS_DESI_FIT_THREADS=2: two groups are fitted side by side (each by its own
    # thread, on its own pair of streams): the latency-bound last rounds of one
    # group's optimiser run under the other's full-size launches
    fit_threads = max(1, int(os.environ.get('RVS_DESI_FIT_THREADS', '1')))
    fpool = concurrent.futures.ThreadPoolExecutor(fit_threads) \
        if fit_threads > 1 else None
    fitting = []    # [(group, future of fit_job)] (fit_threads > 1)
    lanes = list(range(fit_threads))

    def fit_job(fut):
        """a fit thread: wait for the group's preparation, fit it"""
        from .. import vel_fit
        state = fut.result()
        t1 = time.time()
        lane = lanes.pop()
        try:
            with vel_fit.stream_lane(lane):
                res = _group_fit(state, config, kw)
        finally:
            lanes.append(lane)
        return state, res, time.time() - t1

    def finish_oldest_fit():
        group, ff = fitting.pop(0)
        t1 = time.time()
        try:
            state, res, t_fit = ff.result()
        except Exception:  # noqa: BLE001 -- retried per file
            logging.exception('group of %d files failed; retrying one by one'
                              % len(group))
            one_by_one(group)
            return
        GROUP_TIMES['fit'] += time.time() - t1   # (the calling thread's wait)
        GROUP_TIMES['groups'] += 1
        writing.append((group, t_fit, wpool.submit(timed_write, state, res)))
        collect_written()

    def finish_oldest():
        if fpool is not None:
            # hand the oldest prepared group to a fit thread; collect a fit only when
            # more groups are in flight than fit threads
            group, fut = inflight.pop(0)
            fitting.append((group, fpool.submit(fit_job, fut)))
            while len(fitting) > fit_threads:
                finish_oldest_fit()
            return
        group, fut = inflight.pop(0)
        t1 = time.time()
        try:
            from .. import vel_fit
            state = fut.result()
            GROUP_TIMES['wait_prepare'] += time.time() - t1
            t1 = time.time()
            # One worker thread is conditioning the next group meanwhile, another
            # writes the products of the group before; the fit itself runs as
            # vel_fit.process's two halves on two streams (RVS_DESI_SINGLE_STREAM=1:
            # one).  Round 3 measured the split 5 % slower here; with the optimiser's
            # rounds in the library it is 10 % faster (1435 -> 1585 fibres/s).
            import contextlib
            ctx = vel_fit.single_stream() \
                if os.environ.get('RVS_DESI_SINGLE_STREAM') \
                else contextlib.nullcontext()
            with ctx:
                res = _group_fit(state, config, kw)
        except Exception:  # noqa: BLE001 -- retried per file
            logging.exception('group of %d files failed; retrying one by one'
                              % len(group))
            one_by_one(group)
            return
        GROUP_TIMES['fit'] += time.time() - t1
        GROUP_TIMES['groups'] += 1
        writing.append((group, time.time() - t1,
                        wpool.submit(timed_write, state, res)))
        collect_written()

    def flush():
        if not pending:
            return
        group = list(pending)
        del pending[:]
        if len(group) == 1:
            while inflight:
                finish_oldest()
            one_by_one(group)
            return
        inflight.append((group, pool.submit(
            _group_prepare, [(f, t, m, None) for f, t, m in group], config,
            kw_prepare)))
        if len(inflight) > 1:   # group g + 1 is being prepared: fit group g
            finish_oldest()

    def product_names(f):
        """(rvtab, rvmod) paths of input f -- with `subdirs` under the input's
        last two directories -- or None for a path too short for that"""
        parts = f.split('/')
        if subdirs and len(parts) < 3:
            return None
        folder = '/'.join([output_dir] + (parts[-3:-1] if subdirs else [])) + '/'
        os.makedirs(folder, exist_ok=True)
        base = parts[-1][:-3] if parts[-1].endswith('.gz') else parts[-1]
        return tuple('%s%s_%s' % (folder, pre, base)
                     for pre in (output_tab_prefix, output_mod_prefix))

    try:
        # (a shared queue file deals the files itself: utils.FileQueue)
        shared = getattr(files, 'shared', False)
        for f in (files if shared else list(files)[rank::world]):
            names = product_names(f)
            if names is None:
                logging.warning('Invalid file %s: with subdirs it has to be '
                                'dir1/dir2/fname', f)
                continue
            if skipexisting and all(os.path.exists(n) for n in names):
                logging.info('skipping, products already exist %s', f)
                if process_status_file is not None:
                    update_process_status_file(process_status_file, f,
                                               ProcessStatus.EXISTING, -1, 0)
                continue
            pending.append((f, ) + names)
            # (nothing overlaps the preparation of the first group: half a batch)
            limit = files_per_batch // 2 if (files_per_batch >= 4 and
                                             GROUP_TIMES['flushed'] == 0 and
                                             not os.environ.get(
                                                 'RVS_DESI_FULL_FIRST_GROUP')) \
                else max(1, files_per_batch)
            if len(pending) >= limit:
                GROUP_TIMES['flushed'] += 1
                flush()
        flush()
        while inflight:
            finish_oldest()
        while fitting:
            finish_oldest_fit()
    finally:
        # also on the way out of an exception (throw_exceptions): the groups already
        # handed to the writer get their products and status lines, and both
        # worker threads end
        try:
            t1 = time.time()
            collect_written(block=True)
            GROUP_TIMES['drain'] += time.time() - t1
        finally:
            for _, fut in inflight + fitting:
                fut.cancel()
            pool.shutdown()
            if fpool is not None:
                fpool.shutdown()
            wpool.shutdown()
    logging.info('Successfully finished processing')


# ---------------------------------------------------------------------------
# Command line (desi_fit.py:1554-1901, `rvs_desi_fit`): the reference's option names
# with their meaning, so that a survey's job scripts run unchanged --
#   python -m rvspecfit_amd.desi.desi_fit --config config.yaml --output_dir out coadd-*.fits
# What differs, by design: one process drives one GPU (start one process per GPU with
# torch.distributed.run: every rank takes its own stride of the file list, proc_many's
# `shard`, or -- with --queue_file -- its files from the head of the shared queue file,
# utils.FileQueue), so --mpi, the reference's server thread that deals files to CPU
# ranks, is refused with that pointer; --doplot is accepted and ignored.
# ---------------------------------------------------------------------------
_CLI_OPTIONS = (
    # (flags, keyword arguments of add_argument)
    (('input_files', ), dict(nargs='*', default=None, type=str,
                             help='spectra / coadd files to fit')),
    (('--input_file_from', ), dict(type=str, default=None,
                                   help='text file with one input file per line')),
    (('--queue_file', ), dict(action='store_true', default=False,
                              help='--input_file_from is a queue shared by several '
                                   'processes: each takes its files from its head')),
    (('--mpi', ), dict(action='store_true', default=False,
                       help='not supported: start one process per GPU with '
                            'torch.distributed.run instead')),
    (('--nthreads', ), dict(type=int, default=1,
                            help='worker processes sharing this GPU')),
    (('--config', ), dict(type=str, default=None, help='configuration yaml')),
    (('--output_dir', ), dict(type=str, default='./', help='where the products go')),
    (('--output_tab_prefix', ), dict(type=str, default='rvtab',
                                     help='file name prefix of the tables')),
    (('--output_mod_prefix', ), dict(type=str, default='rvmod',
                                     help='file name prefix of the model spectra')),
    (('--targetid', ), dict(type=int, default=None, help='fit this TARGETID only')),
    (('--targetid_file_from', ), dict(type=str, default=None,
                                      help='text file of TARGETIDs to fit')),
    (('--minsn', ), dict(type=float, default=-1e9,
                         help='lowest median S/N (any arm) of a fibre to fit')),
    (('--minexpid', ), dict(type=int, default=None, help='lowest EXPID to fit')),
    (('--maxexpid', ), dict(type=int, default=None, help='highest EXPID to fit')),
    (('--npoly', ), dict(type=int, default=None,
                         help='continuum terms per arm (default 10)')),
    (('--fitarm', ), dict(type=str, default=None,
                          help='arms to fit, comma separated out of b,r,z')),
    (('--figure_dir', ), dict(type=str, default='./', help='(plots are not produced)')),
    (('--figure_prefix', ), dict(type=str, default='fig',
                                 help='(plots are not produced)')),
    (('--log', ), dict(type=str, default=None, help='log file')),
    (('--log_level', ), dict(type=str, default='WARNING',
                             help='DEBUG / INFO / WARNING / ERROR')),
    (('--param_init', ), dict(type=str, default='CCF',
                              help='starting point: CCF or bruteforce')),
    (('--process_status_file', ), dict(type=str, default=None,
                                       help='one status line per processed file')),
    (('--resolution_matrix', ), dict(dest='resolution_matrix', action='store_true',
                                     default=False,
                                     help='fit through the DESI resolution matrices')),
    (('--no-resolution_matrix', ), dict(dest='resolution_matrix',
                                        action='store_false',
                                        help='Gaussian line-spread function of the '
                                             'templates (default)')),
    (('--overwrite', ), dict(default=None, help='(has no meaning any more)')),
    (('--version', ), dict(action='store_true', default=False,
                           help='print the version and leave')),
    (('--skipexisting', ), dict(action='store_true', default=False,
                                help='skip files whose products exist')),
    (('--zbest_select', ), dict(action='store_true', default=False,
                                help='select the fibres to fit by their redrock classification')),
    (('--zbest_include', ), dict(action='store_true', default=False,
                                 help='copy the redrock columns into the table')),
    (('--doplot', ), dict(action='store_true', default=False,
                          help='accepted, ignored: this build makes no plots')),
    (('--no_ccf_continuum_normalize', ), dict(dest='ccf_continuum_normalize',
                                              action='store_false', default=True,
                                              help='cross-correlate without '
                                                   'continuum normalisation')),
    (('--no_subdirs', ), dict(dest='subdirs', action='store_false', default=True,
                              help='no sub-directories below --output_dir')),
    (('--throw_exceptions', ), dict(action='store_true', default=False,
                                    help='let a failing file stop the run')),
    (('--objtypes', ), dict(type=str, default=None,
                            help='comma separated target classes (needs desitarget; '
                                 'ignored without it, as in the reference)')),
    # (not in the reference)
    (('--files_per_batch', ), dict(type=int, default=8,
                                   help='files fitted together in one GPU batch')),
)


def _cli_parser():
    import argparse
    parser = argparse.ArgumentParser(
        prog='rvs_desi_fit',
        description='Radial velocities and stellar parameters of DESI spectra, '
                    'one GPU batch per group of files')
    for flags, kw in _CLI_OPTIONS:
        parser.add_argument(*flags, **kw)
    return parser


def _cli_logging(level, fname):
    lev = getattr(logging, str(level).upper(), None)
    if not isinstance(lev, int):
        raise ValueError('unknown log level %s' % level)
    kw = dict(level=lev, format='%(asctime)s - %(levelname)s - %(message)s',
              force=True)
    if fname is not None:
        kw['filename'] = fname
    logging.basicConfig(**kw)


def main(args=None):
    """desi_fit.main (desi_fit.py:1554-1901): parse the reference's options and run
    proc_many.  Returns nothing; argument errors raise what the reference raises
    (RuntimeError for contradictory inputs, ValueError for unknown arm names or
    --param_init values)."""
    argv = sys.argv[1:] if args is None else list(args)
    cmdline = ' '.join(argv)
    parser = _cli_parser()
    a = parser.parse_args(argv)
    if a.version:
        from .. import __version__ as ver
        print(ver)
        sys.exit(0)
    if a.mpi:
        raise RuntimeError(
            '--mpi deals files to CPU ranks through a server thread; here one process '
            'drives one GPU: start one per GPU with `python -m torch.distributed.run '
            '--nproc-per-node N -m rvspecfit_amd.desi.desi_fit ...` -- every rank fits '
            'its own stride of the file list, or, with --queue_file --input_file_from, '
            'takes its files from the shared queue file')
    if a.queue_file and a.input_file_from is None:
        raise RuntimeError('--queue_file needs --input_file_from (the queue)')
    _cli_logging(a.log_level, a.log)
    fitarm = None
    if a.fitarm is not None:
        fitarm = [x.lower() for x in a.fitarm.split(',')]
        if any(x not in ('b', 'r', 'z') for x in fitarm):
            raise ValueError('only allowed arm names are brz')
    if a.param_init not in ('CCF', 'bruteforce'):
        raise ValueError('Unknown param_init value; only known ones are CCF and '
                         'bruteforce')
    if a.targetid_file_from is not None and a.targetid is not None:
        raise RuntimeError('You can only specify targetid or targetid_file_from '
                           'options')
    fit_targetid = None
    if a.targetid_file_from is not None:
        with open(a.targetid_file_from) as fp:
            fit_targetid = np.unique([int(line) for line in fp if line.strip()])
    elif a.targetid is not None:
        fit_targetid = np.unique([a.targetid])
    files = list(a.input_files or [])
    if files and a.input_file_from is not None:
        raise RuntimeError('You can only specify --input_files OR --input_file_from '
                           'options but not both of them simultaneously')
    if not files and a.input_file_from is None:
        parser.print_help()
        raise RuntimeError('You need to specify the spectra you want to fit')
    if not files:
        files = utils.FileQueue(file_from=a.input_file_from, queue=a.queue_file)
    if a.overwrite is not None:
        logging.warning('overwrite keyword is meaningless now')
    proc_many(files, a.output_dir, a.output_tab_prefix, a.output_mod_prefix,
              figure_dir=a.figure_dir if a.doplot else None,
              figure_prefix=a.figure_prefix, nthreads=a.nthreads,
              config_fname=a.config, fit_targetid=fit_targetid,
              objtypes=None if a.objtypes is None else a.objtypes.split(','),
              doplot=a.doplot, subdirs=a.subdirs, minsn=a.minsn,
              process_status_file=a.process_status_file,
              expid_range=(a.minexpid, a.maxexpid), skipexisting=a.skipexisting,
              fitarm=fitarm, cmdline=cmdline, zbest_select=a.zbest_select,
              zbest_include=a.zbest_include,
              ccf_continuum_normalize=a.ccf_continuum_normalize,
              use_resolution_matrix=a.resolution_matrix,
              ccf_init=(a.param_init == 'CCF'), npoly=a.npoly,
              throw_exceptions=a.throw_exceptions, log_level=a.log_level,
              log_filename=a.log, files_per_batch=max(1, a.files_per_batch))


if __name__ == '__main__':
    main()
