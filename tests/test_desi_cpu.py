"""DESI driver host logic against vectors produced by the
reference's desi/desi_fit.py (tests/golden/make_golden_desi.py), and the FITS
reader/writer against files written by astropy.  No GPU needed."""
import os

import numpy as np
import pytest
import scipy.sparse

from rvspecfit_amd import fits_min as F
from rvspecfit_amd.desi import desi_fit as D

from conftest import GOLD

ARMS = ['b', 'r', 'z']
COADD = os.path.join(GOLD, 'coadd-golden.fits')
SIG0 = dict(b=0.5, r=0.5, z=0.55)


@pytest.fixture(scope='module')
def dcases():
    return dict(np.load(os.path.join(GOLD, 'desi_cases.npz')))


@pytest.fixture(scope='module')
def coadd():
    FP = F.open(COADD, verify_checksum=True)
    return FP, D.read_data(FP, ARMS)


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    if a.dtype.kind in 'SUb':
        return np.array_equal(a, b)
    return np.array_equal(a, b, equal_nan=True)


# ------------------------------------------------------------------ FITS
def test_fits_reader_matches_astropy(dcases, coadd):
    """every image and table column of the coadd as astropy read it, and the
    CHECKSUM/DATASUM of every HDU astropy wrote (verify_checksum above)"""
    FP, _ = coadd
    assert D.valid_file(FP)
    n = 0
    for k, ref in dcases.items():
        if not k.startswith('file/'):
            continue
        parts = k.split('/')
        a = FP[parts[1]].data if len(parts) == 2 else FP[parts[1]].data[parts[2]]
        assert _same(a, ref), k
        assert a.dtype.kind == ref.dtype.kind and \
            a.dtype.itemsize == ref.dtype.itemsize or a.dtype.kind == 'U', k
        n += 1
    assert n > 25
    h = FP[0].header
    assert h['SPGRP'] == 'healpix' and h['HPXPIXEL'] == 10378
    assert h['HPXNEST'] is True


def test_fits_reader_reference_rvtab(dcases):
    """the reference's own RVTAB product: columns, formats, units, comments"""
    T = F.open(os.path.join(GOLD, 'rvtab_ref.fits'), verify_checksum=True)
    assert [h.name for h in T] == list(dcases['plain/tab/extnames'])
    tab = T['RVTAB'].data
    assert tab.columns.names == list(dcases['plain/tab/RVTAB/colnames'])
    assert list(tab.columns.formats) == list(dcases['plain/tab/RVTAB/formats'])
    assert [u or '' for u in tab.columns.units] == \
        list(dcases['plain/tab/RVTAB/units'])
    for i, n in enumerate(tab.columns.names):
        assert _same(tab[n], dcases['plain/tab/RVTAB/col/' + n]), n
        assert T['RVTAB'].header['TCOMM%d' % (i + 1)] == \
            dcases['plain/tab/RVTAB/tcomm'][i]
    keys = list(dcases['plain/tab/primary_keys'])
    for k, v in zip(keys, dcases['plain/tab/primary_vals']):
        assert str(T[0].header[k]) == v, k


def test_fits_writer_roundtrip(tmp_path):
    T = F.open(os.path.join(GOLD, 'rvtab_ref.fits'))
    T[0].header['LONGKEY'] = ('x' * 150 + " it's", 'a comment')
    T[0].header['AFLOAT'] = 1.5e-7
    T[0].header['ABOOL'] = False
    img = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    T.append(F.ImageHDU(img, name='CUBE'))
    T.append(F.ImageHDU(np.arange(7, dtype=np.int64), name='INTS'))
    T.append(F.ImageHDU(None, name='EMPTY'))
    out = str(tmp_path / 'rt.fits')
    T.writeto(out)
    assert os.path.getsize(out) % F.BLOCK == 0
    T2 = F.open(out, verify_checksum=True)  # raises on a bad CHECKSUM/DATASUM
    assert [h.name for h in T2] == [h.name for h in T]
    assert T2[0].header['LONGKEY'] == 'x' * 150 + " it's"
    assert T2[0].header.comment('LONGKEY') == 'a comment'
    assert T2[0].header['AFLOAT'] == 1.5e-7 and T2[0].header['ABOOL'] is False
    assert _same(T2['CUBE'].data, img) and T2['CUBE'].data.dtype == np.float32
    assert _same(T2['INTS'].data, np.arange(7)) and T2['EMPTY'].data is None
    for h, h2 in zip(T, T2):
        if isinstance(h, F.BinTableHDU):
            assert h.data.columns.names == h2.data.columns.names
            assert list(h.data.columns.formats) == list(h2.data.columns.formats)
            for n in h.data.columns.names:
                assert h.data[n].dtype == h2.data[n].dtype
                assert _same(h.data[n], h2.data[n]), n


def test_fits_checksum_detects_corruption(tmp_path):
    raw = bytearray(open(os.path.join(GOLD, 'redrock-golden.fits'), 'rb').read())
    raw[-2000] ^= 0x10
    bad = str(tmp_path / 'bad.fits')
    open(bad, 'wb').write(bytes(raw))
    F.open(bad)
    with pytest.raises(OSError):
        F.open(bad, verify_checksum=True)


# ------------------------------------------------------- fibre selection
def test_get_sns_and_fiberstatus(dcases, coadd):
    FP, (fluxes, ivars, masks, waves, resolutions) = coadd
    for a in ARMS:
        sn = D.get_sns(fluxes[a], ivars[a], masks[a])
        assert _same(sn, dcases['sns/' + a]) and sn.dtype == np.float32
    assert _same(D.fiberstatus_select(FP['FIBERMAP'].data),
                 dcases['fiberstatus_select'])


def test_select_fibers_to_fit(dcases, coadd):
    FP, (fluxes, ivars, masks, waves, resolutions) = coadd
    fm = FP['FIBERMAP'].data
    sns = {a: D.get_sns(fluxes[a], ivars[a], masks[a]) for a in ARMS}
    rrp, rre = D.get_zbest_fname(COADD)
    assert rrp.endswith('redrock-golden.fits') and rre == 'REDSHIFTS'
    assert D.get_zbest_fname('/x/other-golden.fits') == (None, None)
    tid = fm['TARGETID']
    cases = {
        'plain': dict(minsn=2),
        'nosn': dict(minsn=None),
        'tid': dict(minsn=-1e9, fit_targetid=[int(tid[0]), int(tid[8]),
                                              int(tid[7]), 12345]),
        'zinc': dict(minsn=2, zbest_path=rrp, zbest_ext=rre,
                     zbest_include=True),
        'zsel': dict(minsn=2, zbest_path=rrp, zbest_ext=rre, zbest_select=True,
                     objtypes=['MWS_ANY']),
    }
    for k, kw in cases.items():
        sub, rz, rs, rsub = D.select_fibers_to_fit(fm, sns, **kw)
        assert _same(sub, dcases['select/%s/subset' % k]), k
        if rz is None:
            assert 'select/%s/rr_z' % k not in dcases
        else:
            assert _same(rz, dcases['select/%s/rr_z' % k])
            assert _same(rs, dcases['select/%s/rr_spectype' % k])
            assert _same(rsub, dcases['select/%s/rr_subtype' % k])


# ----------------------------------------------------------- conditioning
def test_interpolate_bad_regions(dcases):
    for i in range(8):
        o = D.interpolate_bad_regions(dcases['ibr/%d/spec' % i],
                                      dcases['ibr/%d/mask' % i])
        assert _same(o, dcases['ibr/%d/out' % i]), i


def test_resolution_matrix_helpers(dcases, coadd):
    _, (fluxes, ivars, masks, waves, resolutions) = coadd
    m0 = resolutions['b'][1]
    rows = D.resolution_mat_torows(m0)
    assert _same(rows, dcases['resol/torows'])
    assert _same(D.resolution_mat_tocolumns(rows), dcases['resol/tocolumns'])
    # LAPACK solves: not bit-reproducible across BLAS builds
    dc = D.deconvolve_resolution_matrix(m0, 0.5, 0.8)
    assert np.abs(dc - dcases['resol/deconv']).max() < 1e-13
    sp = D.construct_resolution_sparse_matrix(m0, pix_size_angstrom=0.8,
                                              sigma0_angstrom=0.5)
    assert _same(sp.offsets, dcases['resol/sparse_offsets'])
    assert np.abs(sp.data - dcases['resol/sparse_data']).max() < 1e-13
    assert np.abs(sp @ dcases['resol/vec'] - dcases['resol/sparse_dot']
                  ).max() < 1e-13


@pytest.mark.parametrize('use_res', [False, True])
def test_get_specdata(dcases, coadd, use_res):
    """masking, sigma floor, bad-pixel interpolation, dropped arms: every fibre
    of the synthetic coadd, bit for bit (float32 arithmetic of the reference
    under numpy 1.26); resolution matrices to 1e-13"""
    _, (fluxes, ivars, masks, waves, resolutions) = coadd
    nfib = fluxes['b'].shape[0]
    dropped = 0
    for i in range(nfib):
        with np.errstate(all='ignore'):
            sds = D.get_specdata(waves, fluxes, ivars, masks, resolutions, i,
                                 ARMS, use_resolution_matrix=use_res,
                                 lsf_sigma0_angstrom=SIG0)
        t = 'specdata/%d/%d/' % (int(use_res), i)
        arms = [str(_) for _ in dcases[t + 'arms']]
        assert ([] if sds is None else [s.name for s in sds]) == arms
        dropped += 3 - len(arms)
        for sd in (sds or []):
            for k in ('spec', 'espec', 'badmask'):
                assert _same(getattr(sd, k), dcases[t + sd.name + '/' + k]), \
                    (i, sd.name, k)
            assert sd.spec.dtype == np.float64
            if use_res:
                n = len(sd.spec)
                ref = scipy.sparse.dia_matrix(
                    (dcases[t + sd.name + '/resol_data'],
                     dcases[t + sd.name + '/resol_offsets']),
                    shape=(n, n)).toarray()
                assert np.abs(sd.resolution.mat.toarray() - ref).max() < 1e-13
            else:
                assert sd.resolution is None
    assert dropped >= 5  # fibres 4 (b), 5 (all), 13 (z) exercise the skips


def test_get_specdata_batch_equals_single(coadd):
    _, (fluxes, ivars, masks, waves, resolutions) = coadd
    seq = [0, 3, 4, 5, 10, 13]
    with np.errstate(all='ignore'):
        c = D.get_specdata_batch(waves, fluxes, ivars, masks, resolutions, seq,
                                 ARMS, use_resolution_matrix=True,
                                 lsf_sigma0_angstrom=SIG0)
        for k, i in enumerate(seq):
            sds = D.get_specdata(waves, fluxes, ivars, masks, resolutions, i,
                                 ARMS, use_resolution_matrix=True,
                                 lsf_sigma0_angstrom=SIG0)
            names = [] if sds is None else [s.name for s in sds]
            for a in ARMS:
                assert bool(c[a]['ok'][k]) == ('desi_' + a in names)
            for sd in (sds or []):
                a = sd.name[-1]
                assert _same(c[a]['spec'][k], sd.spec)
                assert _same(c[a]['espec'][k], sd.espec)
                assert _same(c[a]['badmask'][k], sd.badmask)


# ------------------------------------------------------- bits and schema
def test_rvs_warn_bits(dcases):
    cfg = dict(min_vel=-1000, max_vel=1000)
    w = D.rvs_warn_bits(dcases['warn/CHISQ_TOT'], dcases['warn/CHISQ_C_TOT'],
                        dcases['warn/VRAD'], dcases['warn/VSINI'],
                        dcases['warn/VRAD_ERR'], dcases['warn/bad_hessian'],
                        dcases['warn/teff'], dcases['warn/feh'],
                        dcases['warn/logg'], cfg)
    assert _same(w, dcases['warn/warn'])
    assert set(np.unique(w)) >= {0, 1, 2, 4, 8, 16, 64}
    for i in range(len(w)):
        od = dict(CHISQ_TOT=dcases['warn/CHISQ_TOT'][i],
                  CHISQ_C_TOT=dcases['warn/CHISQ_C_TOT'][i],
                  VRAD=dcases['warn/VRAD'][i], VSINI=dcases['warn/VSINI'][i],
                  VRAD_ERR=dcases['warn/VRAD_ERR'][i])
        fr = dict(bad_hessian=bool(dcases['warn/bad_hessian'][i]),
                  param=dict(teff=dcases['warn/teff'][i],
                             feh=dcases['warn/feh'][i],
                             logg=dcases['warn/logg'][i]))
        assert D.get_rvs_warn(fr, od, cfg) == w[i]
    assert D.bitmasks['BAD_SPECTRUM'] == 32


def test_column_desc(dcases):
    cd = D.get_column_desc(ARMS)
    assert list(cd.keys()) == list(dcases['coldesc/names'])
    assert [v[1] for v in cd.values()] == list(dcases['coldesc/comments'])


def test_rows_to_table_missing_cells():
    rows = [dict(A=1.5, N=3, S='ab', B=True), dict(N=4, C=2.0, S='abcd'),
            dict(A=2.5)]
    t = D.rows_to_table(rows)
    assert t.columns.names == ['A', 'N', 'S', 'B', 'C']
    assert _same(t['A'], [1.5, np.nan, 2.5])
    assert _same(t['N'], [3, 4, D.INT_NULL]) and t.column('N').null == 999999
    assert list(t['S']) == ['ab', 'abcd', '']
    assert _same(t['C'], [np.nan, 2.0, np.nan])


# ----------------------------------------- proc_desi paths without a fit
CFG = dict(template_lib='golden-desi://', min_vel=-1000, max_vel=1000,
           min_vel_step=0.2, vel_step0=5, min_vsini=0.1, max_vsini=500,
           second_minimizer=True, config_file_path='/x/config.yaml',
           lsf_sigma0_angstrom=SIG0)


def test_proc_desi_nothing_selected(dcases, tmp_path):
    """minsn above every fibre: RVTAB without rows, MODEL images without data"""
    tab, mod = str(tmp_path / 't.fits'), str(tmp_path / 'm.fits')
    n = D.proc_desi(COADD, tab, mod, None, CFG, doplot=False, minsn=1e9,
                    cmdline='golden none')
    assert n == 0 == int(dcases['none/nfit'])
    T, M = F.open(tab, verify_checksum=True), F.open(mod, verify_checksum=True)
    assert [h.name for h in T] == list(dcases['none/tab/extnames'])
    assert [h.name for h in M] == list(dcases['none/mod/extnames'])
    assert len(T['RVTAB'].data) == 0 and len(T['FIBERMAP'].data) == 0
    assert M['B_MODEL'].data is None
    assert _same(M['R_WAVELENGTH'].data, dcases['none/mod/R_WAVELENGTH'])
    keys = list(dcases['none/mod/primary_keys'])
    for k, v in zip(keys, dcases['none/mod/primary_vals']):
        if k not in ('RVS_CONF', ):
            assert str(M[0].header[k]) == v, k


def test_proc_desi_unknown_targetid(dcases, tmp_path):
    tab, mod = str(tmp_path / 't.fits'), str(tmp_path / 'm.fits')
    n = D.proc_desi(COADD, tab, mod, None, CFG, doplot=False,
                    fit_targetid=[77])
    assert n == 0 == int(dcases['notid/nfit'])
    assert [h.name for h in F.open(tab)] == list(dcases['notid/tab/extnames'])
    assert [h.name for h in F.open(mod)] == ['PRIMARY']


def test_proc_desi_invalid_inputs(tmp_path):
    assert D.proc_desi(str(tmp_path / 'nope.fits'), 'a', 'b', None, CFG) == -1
    rr = os.path.join(GOLD, 'redrock-golden.fits')  # readable, wrong layout
    assert D.proc_desi(rr, 'a', 'b', None, CFG) == -1


def test_proc_many_status_file(tmp_path):
    """the wrapper's status file and the rank stride of the file list"""
    import yaml
    cfgf = str(tmp_path / 'c.yaml')
    with open(cfgf, 'w') as fp:
        yaml.safe_dump(dict(template_lib='golden-desi://'), fp)
    st = str(tmp_path / 'status')
    D.proc_many([COADD, str(tmp_path / 'x/y/missing.fits')], str(tmp_path),
                'rvtab', 'rvmod', config_fname=cfgf, minsn=1e9, doplot=False,
                subdirs=False, process_status_file=st, shard=(0, 1))
    lines = open(st).read().split('\n')
    assert lines[0].split()[:3] == [COADD, 'SUCCESS', '0']
    assert lines[1].split()[1] == 'FAILURE'
    assert os.path.exists(str(tmp_path / 'rvtab_coadd-golden.fits'))
    st2 = str(tmp_path / 'status2')
    D.proc_many([COADD, COADD, COADD], str(tmp_path), 'rvtab', 'rvmod',
                config_fname=cfgf, minsn=1e9, doplot=False, subdirs=False,
                process_status_file=st2, shard=(1, 2), skipexisting=True)
    rows = open(st2 + '.1').read().strip().split('\n')
    assert len(rows) == 1 and rows[0].split()[1] == 'EXISTING'


def test_proc_many_writer_pipeline_drains_and_redoes_only_failed_files(
        tmp_path, monkeypatch):
    """proc_many's prepare / fit / write pipeline with the GPU part replaced: a file
    whose product assembly raises is redone ALONE (the others of its group keep
    their products and status lines), and an exception that leaves the file loop
    (throw_exceptions) still lets the groups already handed to the writer finish
    and ends both worker threads"""
    import threading
    import yaml
    cfgf = str(tmp_path / 'c.yaml')
    with open(cfgf, 'w') as fp:
        yaml.safe_dump(dict(template_lib='golden-desi://'), fp)
    files = [str(tmp_path / ('f%d.fits' % i)) for i in range(7)]
    written, redone = [], []

    def steps(fname, tab, mod, fig, config, **kw):
        res = yield ('request', fname)
        if fname.endswith('f2.fits'):
            raise IOError('disk full')
        written.append((fname, res))
        return 3

    def fit(state, config, kw):
        if any(r[1].endswith('f5.fits') for _, _, r in state[0]):
            raise RuntimeError('fit blew up')
        return ['fit:' + r[1] for _, _, r in state[0]]

    def wrapper(f, t, m, fig, config, process_status_file=None,
                throw_exceptions=None, **kw):
        redone.append(f)
        if throw_exceptions:
            raise RuntimeError('escapes the file loop')

    monkeypatch.setattr(D, '_proc_desi_steps', steps)
    monkeypatch.setattr(D, '_group_fit', fit)
    monkeypatch.setattr(D, 'proc_desi_wrapper', wrapper)
    st = str(tmp_path / 'status')
    kw = dict(config_fname=cfgf, doplot=False, subdirs=False, shard=(0, 1),
              files_per_batch=2)
    n0 = threading.active_count()
    D.proc_many(files[:4], str(tmp_path), 'rvtab', 'rvmod',
                process_status_file=st, **kw)
    assert redone == [files[2]]                      # only the file that failed
    assert [w[0] for w in written] == [files[0], files[1], files[3]]
    assert all(w[1] == 'fit:' + w[0] for w in written)
    rows = [l.split() for l in open(st).read().strip().split('\n')]
    assert [r[0] for r in rows if r[1] == 'SUCCESS'] == [files[0], files[1],
                                                         files[3]]
    assert threading.active_count() == n0            # both pools shut down
    # groups (0,1) (2,3) (4,5) (6): the fit of (4,5) raises, its one-by-one retry
    # raises out of proc_many -- (0,1) and (2,3) were handed to the writer before
    del written[:], redone[:]
    with pytest.raises(RuntimeError, match='escapes'):
        D.proc_many(files, str(tmp_path), 'rvtab', 'rvmod', throw_exceptions=True,
                    process_status_file=st + '2', **kw)
    assert [w[0] for w in written] == [files[0], files[1], files[3]]
    assert threading.active_count() == n0
    rows = [l.split() for l in open(st + '2').read().strip().split('\n')]
    assert [r[0] for r in rows if r[1] == 'SUCCESS'] == [files[0], files[1],
                                                         files[3]]


def test_proc_many_device_selection(tmp_path, monkeypatch):
    """one process per GPU: the fitting process takes the GPU LOCAL_RANK names
    whatever its file shard is -- the worker processes of `nthreads` run with
    shard (0, 1) and must not all land on GPU 0 --, and a parent that only
    waits for its workers never selects (touches) a GPU"""
    import torch
    import yaml
    calls = []
    monkeypatch.setattr(torch.cuda, 'set_device', lambda i: calls.append(i))
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 8)
    assert D._select_rank_device({}) is None and calls == []
    assert D._select_rank_device(dict(LOCAL_RANK='5')) == 5 and calls == [5]
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 4)
    # more ranks than GPUs is refused unless the caller opts in (bench.py's rule)
    with pytest.raises(RuntimeError):
        D._select_rank_device(dict(LOCAL_RANK='5'))
    assert D._select_rank_device(dict(LOCAL_RANK='5', RVS_SHARE_GPU='1')) == 1
    del calls[:]
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 8)
    cfgf = str(tmp_path / 'c.yaml')
    with open(cfgf, 'w') as fp:
        yaml.safe_dump(dict(template_lib='golden-desi://'), fp)
    kw = dict(config_fname=cfgf, minsn=1e9, doplot=False, subdirs=False)
    # a worker as proc_many spawns it: shard (0, 1), the rank's environment
    monkeypatch.setenv('LOCAL_RANK', '3')
    monkeypatch.setenv('RANK', '3')
    monkeypatch.setenv('WORLD_SIZE', '8')
    D.proc_many([], str(tmp_path), 'rvtab', 'rvmod', shard=(0, 1), nthreads=1,
                **kw)
    assert calls == [3]
    # a rank of the launcher
    del calls[:]
    D.proc_many([], str(tmp_path), 'rvtab', 'rvmod', **kw)
    assert calls == [3]
    # the parent of worker processes: no device selected here
    del calls[:]
    D.proc_many([], str(tmp_path), 'rvtab', 'rvmod', nthreads=2, **kw)
    assert calls == []
    # no launcher: the current device stays
    monkeypatch.delenv('LOCAL_RANK')
    D.proc_many([], str(tmp_path), 'rvtab', 'rvmod', shard=(0, 1), **kw)
    assert calls == []


def test_select_expid_range_and_id_matched_redshifts(coadd, tmp_path):
    """spectra- files: the EXPID window (desi_fit.py:590-597; open ends, and no
    window at all -- where the reference trips over an unset variable -- select
    everything), and a redshift table with a DIFFERENT row count, matched by
    TARGETID (desi_fit.py:655-672)"""
    FP, (fluxes, ivars, masks, waves, resolutions) = coadd
    fm0 = FP['FIBERMAP'].data
    n = len(fm0)
    fm = F.FitsTable([F.Column(c.name, c.array, c.unit, c.tform)
                      for c in fm0._cols])
    fm.add('EXPID', np.arange(n, dtype=np.int32))
    sns = {a: np.full(n, 10.0) for a in ARMS}
    base = D.select_fibers_to_fit(fm, sns)[0]
    assert base.sum() == n - 2  # the SKY fibre and the bad FIBERSTATUS one
    sub = D.select_fibers_to_fit(fm, sns, expid_range=(5, 8))[0]
    assert np.array_equal(np.nonzero(sub)[0], [8])  # (5, 8] minus fibres 6, 7
    sub = D.select_fibers_to_fit(fm, sns, expid_range=(None, 3))[0]
    assert np.array_equal(np.nonzero(sub)[0], [0, 1, 2, 3])
    sub = D.select_fibers_to_fit(fm, sns, expid_range=(10, None))[0]
    assert np.array_equal(np.nonzero(sub)[0], [11, 12, 13])
    # redshift table with half the rows, in another order
    rr0 = F.open(os.path.join(GOLD, 'redrock-golden.fits'))['REDSHIFTS'].data
    keep = np.array([9, 3, 0, 12, 8, 1])
    rr = F.FitsTable([F.Column(c.name, c.array[keep], c.unit)
                      for c in rr0._cols])
    path = str(tmp_path / 'redrock-part.fits')
    F.HDUList([F.PrimaryHDU(), F.BinTableHDU(rr, name='REDSHIFTS')]).writeto(path)
    sub, rz, rs, rsub = D.select_fibers_to_fit(
        fm, sns, zbest_path=path, zbest_ext='REDSHIFTS', zbest_select=True)
    # stars (or |z| small) among the listed ones: all but the galaxy (fibre 3)
    assert np.array_equal(np.nonzero(sub)[0], [0, 1, 8, 9, 12])
    assert np.isnan(rz[2]) and rs[2] == '' and rz[9] == rr0['Z'][9]
    assert rs[3] == 'GALAXY' and rsub[0] == 'K'


def test_command_line_arguments():
    """desi_fit.main's front end (desi_fit.py:1554-1901): the reference's option
    names, defaults and argument errors; nothing is fitted (no GPU)"""
    from rvspecfit_amd.desi import desi_fit as D
    a = D._cli_parser().parse_args(['x.fits', 'y.fits'])
    assert a.input_files == ['x.fits', 'y.fits'] and a.input_file_from is None
    assert (a.nthreads, a.output_dir, a.output_tab_prefix, a.output_mod_prefix) == \
        (1, './', 'rvtab', 'rvmod')
    assert a.minsn == -1e9 and a.npoly is None and a.fitarm is None
    assert a.param_init == 'CCF' and a.resolution_matrix is False
    assert a.ccf_continuum_normalize is True and a.subdirs is True
    assert not (a.skipexisting or a.zbest_select or a.zbest_include or a.doplot
                or a.throw_exceptions or a.mpi or a.queue_file)
    b = D._cli_parser().parse_args(['x.fits', '--resolution_matrix',
                                    '--no_ccf_continuum_normalize', '--no_subdirs',
                                    '--minexpid', '3', '--maxexpid', '9'])
    assert b.resolution_matrix and not b.ccf_continuum_normalize and not b.subdirs
    assert (b.minexpid, b.maxexpid) == (3, 9)
    assert not D._cli_parser().parse_args(
        ['x.fits', '--resolution_matrix', '--no-resolution_matrix']).resolution_matrix
    with pytest.raises(RuntimeError, match='specify the spectra'):
        D.main([])
    with pytest.raises(RuntimeError, match='not both'):
        D.main(['x.fits', '--input_file_from', 'list.txt'])
    with pytest.raises(RuntimeError, match='targetid or targetid_file_from'):
        D.main(['x.fits', '--targetid', '5', '--targetid_file_from', 'ids.txt'])
    with pytest.raises(ValueError, match='arm names'):
        D.main(['x.fits', '--fitarm', 'b,q'])
    with pytest.raises(ValueError, match='param_init'):
        D.main(['x.fits', '--param_init', 'guess'])
    with pytest.raises(RuntimeError, match='torch.distributed.run'):
        D.main(['x.fits', '--mpi'])
    with pytest.raises(RuntimeError, match='needs --input_file_from'):
        D.main(['x.fits', '--queue_file'])
    with pytest.raises(SystemExit):
        D.main(['--version'])


def _take_all(path, out):
    from rvspecfit_amd import utils
    out.put(list(utils.FileQueue(file_from=path, queue=True, wait=(0.001, 0.003))))


def test_file_queue_shared_by_processes(tmp_path):
    """utils.FileQueue (utils.py:113-177): a list, a text file read once, and the text
    file as a queue that three processes empty together -- every line goes to exactly
    one of them, the file is left empty (the rename-lock protocol of the reference, so
    its CPU workers and this build's GPU processes can share one queue file)"""
    import multiprocessing as mp
    from rvspecfit_amd import utils
    names = ['coadd-%03d.fits' % i for i in range(60)]
    q = str(tmp_path / 'queue.txt')
    with open(q, 'w') as fp:
        fp.write(''.join(n + '\n' for n in names))
    assert list(utils.FileQueue(file_list=names[:3])) == names[:3]
    once = utils.FileQueue(file_from=q)
    assert not once.shared and list(once) == names
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    procs = [ctx.Process(target=_take_all, args=(q, out)) for _ in range(3)]
    for p_ in procs:
        p_.start()
    got = [out.get(timeout=120) for _ in procs]
    for p_ in procs:
        p_.join()
    assert sorted(sum(got, [])) == names
    assert open(q).read() == ''
    fq = utils.FileQueue(file_from=q, queue=True)
    assert fq.shared and list(fq) == []
    with pytest.raises(ValueError):
        utils.FileQueue()
