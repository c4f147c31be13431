"""End-to-end DESI driver on the GPU: proc_desi on the
synthetic coadd against the RVTAB / RVMOD the reference's own proc_desi wrote
for the same file (tests/golden/make_golden_desi.py)."""
import os

import numpy as np
import pytest

from conftest import GOLD

pytestmark = pytest.mark.gpu

COADD = os.path.join(GOLD, 'coadd-golden.fits')
SIG0 = dict(b=0.5, r=0.5, z=0.55)
CFG = dict(template_lib='golden-desi://', min_vel=-1000, max_vel=1000,
           min_vel_step=0.2, vel_step0=5, min_vsini=0.1, max_vsini=500,
           second_minimizer=True, config_file_path='/x/config.yaml',
           lsf_sigma0_angstrom=SIG0)
# not pinned by the real numdifftools (make_golden_desi.py header)
ERR_COLS = ('LOGG_ERR', 'TEFF_ERR', 'FEH_ERR', 'ALPHAFE_ERR')


@pytest.fixture(scope='module')
def dcases():
    return dict(np.load(os.path.join(GOLD, 'desi_cases.npz')))


@pytest.fixture(scope='module')
def desi_libs():
    from rvspecfit_amd import spec_inter
    from rvspecfit_amd.library import TemplateLibrary
    for n in ('desi_b', 'desi_r', 'desi_z'):
        lib = TemplateLibrary(n, np.load(os.path.join(GOLD, 'lib_%s.npz' % n)))
        spec_inter.register_library(lib, 'golden-desi://')
    return True


def _tid(idx):
    from rvspecfit_amd import fits_min as F
    t = F.open(COADD)['FIBERMAP'].data['TARGETID']
    return [int(t[_]) for _ in idx]


RUNS = {
    'plain': lambda: dict(minsn=2, zbest_include=True),
    'resol': lambda: dict(minsn=2, use_resolution_matrix=True,
                          fit_targetid=_tid((0, 2, 4, 11))),
    'armb': lambda: dict(minsn=2, fitarm=['b', 'z'],
                         fit_targetid=_tid((1, 4, 12))),
    'noccf': lambda: dict(minsn=2, ccf_init=False, fit_targetid=_tid((0, 8))),
    'bad': lambda: dict(minsn=None, fit_targetid=_tid((4, 5))),
}


@pytest.mark.parametrize('tag', list(RUNS))
def test_proc_desi_against_reference(dcases, desi_libs, tmp_path, tag):
    from rvspecfit_amd import fits_min as F
    from rvspecfit_amd.desi import desi_fit as D
    tabf, modf = str(tmp_path / 'rvtab.fits'), str(tmp_path / 'rvmod.fits')
    n = D.proc_desi(COADD, tabf, modf, None, CFG, doplot=False,
                    cmdline='golden ' + tag, **RUNS[tag]())
    assert n == int(dcases[tag + '/nfit'])
    T = F.open(tabf, verify_checksum=True)
    M = F.open(modf, verify_checksum=True)
    p = tag + '/tab/RVTAB/'
    # ---- schema: HDUs, columns, formats, units, comments, header keywords
    assert [h.name for h in T] == list(dcases[tag + '/tab/extnames'])
    assert [h.name for h in M] == list(dcases[tag + '/mod/extnames'])
    tab = T['RVTAB'].data
    assert tab.columns.names == list(dcases[p + 'colnames'])
    assert list(tab.columns.formats) == list(dcases[p + 'formats'])
    assert [u or '' for u in tab.columns.units] == list(dcases[p + 'units'])
    for i in range(len(tab.columns.names)):
        assert T['RVTAB'].header['TCOMM%d' % (i + 1)] == dcases[p + 'tcomm'][i]
    for which, H in (('tab', T), ('mod', M)):
        for k, v in zip(dcases['%s/%s/primary_keys' % (tag, which)],
                        dcases['%s/%s/primary_vals' % (tag, which)]):
            if k in ('RVS_CONF', 'RR_FILE'):
                assert k in H[0].header
            elif k.startswith('TMPL'):
                continue  # the interpolator cache of the process, below
            else:
                assert str(H[0].header[k]) == v, (which, k)
    if n > 0:
        cons = [v for k, v in T[0].header.items() if k.startswith('TMPLCON')]
        assert set(cons) <= {'desi_b', 'desi_r', 'desi_z'} and len(cons) >= 2
    assert np.array_equal(T['FIBERMAP'].data['TARGETID'],
                          dcases[tag + '/tab/FIBERMAP/TARGETID'])
    assert np.array_equal(T['SCORES'].data['TARGETID'],
                          dcases[tag + '/tab/SCORES/TARGETID'])
    # ---- values
    ref = {c: dcases[p + 'col/' + c] for c in tab.columns.names}
    chi, chi_ref = tab['CHISQ_TOT'], ref['CHISQ_TOT']
    good = np.isfinite(chi_ref)
    assert np.array_equal(np.isfinite(chi), good)
    # same optimum (to the optimiser's own tolerance) or a better one.  Nelder-
    # Mead stops when its simplex spans < fatol = 1e-3 in chi^2, and a last-bit
    # difference of one objective value sends it down another (equally valid)
    # path: with the nearest-neighbour query as p / ptp (cKDTree's form) instead
    # of p * (1 / ptp) -- 1e-16 of the outside penalty -- the single-arm case
    # ends 3.0e-3 above the reference's end point, with the reciprocal 2e-4 below
    assert np.all(chi[good] <= chi_ref[good] + 5e-3)
    same = good & (np.abs(chi - chi_ref) <= 3e-7 * np.abs(chi_ref))
    assert same.sum() >= 1
    exact = ('TARGETID', 'FIBER', 'REF_ID', 'REF_CAT', 'TARGET_RA',
             'TARGET_DEC', 'NPIX_TOT', 'RR_Z', 'RR_SPECTYPE', 'RR_SUBTYPE')
    for c in tab.columns.names:
        a, b = tab[c], ref[c]
        if c in exact or c.startswith('SN_'):
            assert np.array_equal(a, b) if a.dtype.kind in 'SUb' else \
                np.array_equal(a, b, equal_nan=True), c
        elif c in ('RVS_WARN', 'SUCCESS'):
            continue
        elif c in ERR_COLS:
            # numdifftools' Hessian as restated (numdiff.py), at this run's own
            # optimum: where that is the reference's, finite differences of a
            # 1e-9-noisy function agree to a few per cent (at the reference's
            # own end point: 1e-3, test_param_uncertainties_at_reference_optimum)
            assert np.array_equal(np.isnan(a), np.isnan(b)), c
            ok = same & np.isfinite(b) & (b > 0)
            assert np.allclose(a[ok], b[ok], rtol=5e-2), c
        elif c.startswith('CHISQ_C'):
            assert np.allclose(a, b, rtol=1e-9, atol=0, equal_nan=True), c
        elif c in ('VRAD', 'VRAD_CCF'):
            # north_star: RV within 0.01 km/s -- also at a different optimum
            assert np.nanmax(np.abs(a - b)) <= 0.01, c
            assert np.nanmax(np.abs(a[same] - b[same])) <= 1e-3, c
        elif c.startswith('CHISQ_'):
            assert np.allclose(a[same], b[same], rtol=1e-4, equal_nan=True), c
        elif c in ('VRAD_SKEW', 'VRAD_KURT', 'VRAD_ERR'):
            assert np.allclose(a[same], b[same], rtol=2e-2, atol=2e-3), c
        else:  # VSINI, LOGG, TEFF, FEH, ALPHAFE: where the optimum is the same
            scale = dict(TEFF=1.0, VSINI=0.05).get(c, 2e-3)
            assert np.nanmax(np.abs(a[same] - b[same])) <= scale, c
    # the warning bits: identical, BAD_HESSIAN included, where the optimiser
    # ended in the reference's optimum; elsewhere every bit but that one
    bh = D.bitmasks['BAD_HESSIAN']
    w, w_ref = tab['RVS_WARN'], ref['RVS_WARN']
    assert np.array_equal(w[same], w_ref[same])
    assert np.array_equal(w & ~bh, w_ref & ~bh)
    assert np.array_equal(tab['SUCCESS'], w == 0)
    # ---- models (float32 images)
    for h in M[1:]:
        if h.name.endswith('_WAVELENGTH'):
            assert np.array_equal(h.data, dcases['%s/mod/%s' % (tag, h.name)])
        elif h.name.endswith('_MODEL'):
            b = dcases['%s/mod/%s' % (tag, h.name)]
            assert h.data.shape == b.shape and h.data.dtype == np.float32
            rows = np.nonzero(same[:len(b)])[0]
            # (masked stretches leave the continuum polynomial nearly free:
            # the model there follows the last digits of the parameters)
            sc = np.abs(b).max()
            assert np.nanmax(np.abs(h.data[rows] - b[rows])) <= 3e-4 * sc
            good_px = np.abs(h.data[rows] - b[rows]) <= 2e-5 * sc
            assert good_px.mean() > 0.95


def test_bad_spectrum_row(dcases, desi_libs, tmp_path):
    """the all-masked fibre: BAD_SPECTRUM, empty cells, no model row"""
    from rvspecfit_amd import fits_min as F
    from rvspecfit_amd.desi import desi_fit as D
    tabf, modf = str(tmp_path / 't.fits'), str(tmp_path / 'm.fits')
    D.proc_desi(COADD, tabf, modf, None, CFG, doplot=False, **RUNS['bad']())
    tab = F.open(tabf)['RVTAB'].data
    assert list(tab['RVS_WARN'])[1] == D.bitmasks['BAD_SPECTRUM']
    assert np.isnan(tab['VRAD'][1]) and tab['NPIX_TOT'][1] == D.INT_NULL
    assert not tab['SUCCESS'][1]
    assert F.open(modf)['R_MODEL'].data.shape[0] == 1
    # a bad fibre AHEAD of a good one (the reference raises IndexError there):
    # model rows follow the table rows
    D.proc_desi(COADD, tabf, modf, None, CFG, doplot=False, minsn=None,
                fit_targetid=_tid((5, 10)))
    tab = F.open(tabf)['RVTAB'].data
    assert list(tab['RVS_WARN'] & 32) == [32, 0]
    assert tab.columns.names[0] == 'RVS_WARN'
    m = F.open(modf)['R_MODEL'].data
    assert m.shape[0] == 2 and not m[0].any() and m[1].any()


def test_proc_onespec_equals_batch(dcases, desi_libs):
    """the per-fibre entry point (one SpecData tuple) and the batched file
    path are the same code: identical numbers"""
    from rvspecfit_amd import fits_min as F
    from rvspecfit_amd.desi import desi_fit as D
    FP = F.open(COADD)
    fl, iv, ms, wv, rs = D.read_data(FP, ['b', 'r', 'z'])
    sds = D.get_specdata(wv, fl, iv, ms, rs, 8, ['b', 'r', 'z'])
    od, yfit = D.proc_onespec(sds, ['b', 'r', 'z'], CFG, dict(npoly=10),
                              doplot=False)
    k = list(dcases['plain/tab/RVTAB/col/TARGETID']).index(_tid((8, ))[0])
    p = 'plain/tab/RVTAB/col/'
    assert abs(od['VRAD'] - dcases[p + 'VRAD'][k]) < 1e-3
    assert abs(od['CHISQ_TOT'] - dcases[p + 'CHISQ_TOT'][k]) < 1e-4 * \
        dcases[p + 'CHISQ_TOT'][k]
    assert od['RVS_WARN'] == dcases[p + 'RVS_WARN'][k]
    assert set(od['versions']) >= {'desi_b', 'desi_r', 'desi_z'}
    assert len(yfit) == 3 and yfit[0].shape == wv['b'].shape


def test_process_device_nm_uses_resolution(desi_libs):
    """the objective of the device Nelder-Mead applies the spectra's resolution
    matrices (per-spectrum taps): the kernel chain (the path resolution matrices
    take) reaches the optimum the torch reference machine reaches with the
    batched get_chisq as objective, and a different one from the fit without
    matrices"""
    import torch
    from refmachines import neldermead_torch
    from rvspecfit_amd import fits_min as F, vel_fit, engine, optimizer
    from rvspecfit_amd.desi import desi_fit as D
    FP = F.open(COADD)
    data = D.read_data(FP, ['b', 'r', 'z'])
    fl, iv, ms, wv, rs = data
    cond = D.get_specdata_batch(wv, fl, iv, ms, rs, [0, 2, 11], ['b', 'r', 'z'],
                                use_resolution_matrix=True,
                                lsf_sigma0_angstrom=SIG0)
    batch = D._arm_batch(cond, ['b', 'r', 'z'], (True, True, True),
                         np.arange(3), wv, 'cuda')
    p0 = dict(teff=5200., logg=2.5, feh=-1., alpha=0.2, vsini=10.)
    cfg = dict(CFG, second_minimizer=False)
    a = vel_fit.process(batch, dict(p0), config=cfg, options=dict(npoly=10))

    # the same simplices driven by the torch machine over vel_fit._Objective
    # (spec_fit.chisq_jobs: rvs_chisq_point with the taps)
    class TorchNM(optimizer.DeviceNelderMead):
        def minimize(self, pobj, simplex, **kw):
            names = ['teff', 'logg', 'feh', 'alpha']
            pd = vel_fit._as_param_tensors(p0, batch.S, batch.device)
            mapper = vel_fit.ParamMapper(names, pd, [],
                                         vel_fit.VSiniMapper(cfg['max_vsini']),
                                         fitVsini=True)
            obj = vel_fit._Objective(batch, mapper, cfg, dict(npoly=10), None)
            obj.safe_params = torch.stack([pd[_] for _ in names], dim=1)
            return neldermead_torch.minimize(obj, simplex, **kw)
    keep = optimizer.DeviceNelderMead
    optimizer.DeviceNelderMead = TorchNM
    try:
        b = vel_fit.process(batch, dict(p0), config=cfg, options=dict(npoly=10))
    finally:
        optimizer.DeviceNelderMead = keep
    assert np.allclose(a['chisq'].cpu().numpy(), b['chisq'].cpu().numpy(),
                       rtol=1e-7)
    assert np.allclose(a['vel'].cpu().numpy(), b['vel'].cpu().numpy(),
                       atol=1e-3)
    assert np.array_equal(a['nm_nit'].cpu().numpy(), b['nm_nit'].cpu().numpy())
    for arm in batch.arms:
        arm.resol = None
    c = vel_fit.process(engine.SpecBatch(batch.arms), dict(p0), config=cfg,
                        options=dict(npoly=10))
    assert not np.allclose(a['chisq'].cpu().numpy(), c['chisq'].cpu().numpy(),
                           rtol=1e-4)


def test_grouped_files_equal_single(dcases, desi_libs, tmp_path):
    """several files fitted together (proc_desi_group, what proc_many does):
    every fibre's numbers are those of the file processed alone, bit for bit --
    the lock-step optimiser, the CCF and the chi^2 kernels do not let one
    spectrum see another"""
    from rvspecfit_amd import fits_min as F
    from rvspecfit_amd.desi import desi_fit as D
    kw = dict(doplot=False, minsn=2, zbest_include=True)
    single = (str(tmp_path / 's_tab.fits'), str(tmp_path / 's_mod.fits'))
    n1 = D.proc_desi(COADD, single[0], single[1], None, CFG, **kw)
    files = [(COADD, str(tmp_path / ('g%d_tab.fits' % i)),
              str(tmp_path / ('g%d_mod.fits' % i)), None) for i in range(3)]
    # the third member selects other fibres: a different mix in the batch
    rets = D.proc_desi_group(files[:2], CFG, **kw)
    assert rets == [n1, n1]
    ts = F.open(single[0])['RVTAB'].data
    ms = F.open(single[1])
    for _, tab, mod, _ in files[:2]:
        tg = F.open(tab)['RVTAB'].data
        assert tg.columns.names == ts.columns.names
        for c in ts.columns.names:
            a, b = ts[c], tg[c]
            assert np.array_equal(a, b) if a.dtype.kind in 'SUb' else \
                np.array_equal(a, b, equal_nan=True), c
        mg = F.open(mod)
        for h in ms[1:]:
            if h.name.endswith('_MODEL'):
                assert np.array_equal(h.data, mg[h.name].data), h.name
    # proc_many drives the same path and keeps the status file per input file
    import yaml
    cfgf = str(tmp_path / 'c.yaml')
    with open(cfgf, 'w') as fp:
        yaml.safe_dump({k: v for k, v in CFG.items()
                        if k != 'config_file_path'}, fp)
    link = str(tmp_path / 'coadd-copy.fits')
    os.symlink(COADD, link)
    os.symlink(os.path.join(GOLD, 'redrock-golden.fits'),
               str(tmp_path / 'redrock-copy.fits'))
    st = str(tmp_path / 'status')
    D.proc_many([COADD, link], str(tmp_path / 'out'), 'rvtab', 'rvmod',
                config_fname=cfgf, minsn=2, zbest_include=True, doplot=False,
                subdirs=False, process_status_file=st, shard=(0, 1),
                files_per_batch=2)
    rows = [l.split() for l in open(st).read().strip().split('\n')]
    assert [r[1] for r in rows] == ['SUCCESS', 'SUCCESS']
    assert [int(r[2]) for r in rows] == [n1, n1]
    tm = F.open(str(tmp_path / 'out' / 'rvtab_coadd-copy.fits'))['RVTAB'].data
    assert np.array_equal(tm['VRAD'], ts['VRAD'])


def test_command_line_runs_the_file_loop(dcases, desi_libs, tmp_path):
    """desi_fit.main (desi_fit.py:1554-1901) with the reference's option names: two
    files from --input_file_from, the products of proc_desi on the file alone, the
    command line in the headers' RVS_CMD, the status file"""
    import yaml
    from rvspecfit_amd import fits_min as F
    from rvspecfit_amd.desi import desi_fit as D
    single = (str(tmp_path / 's_tab.fits'), str(tmp_path / 's_mod.fits'))
    n1 = D.proc_desi(COADD, single[0], single[1], None, CFG, doplot=False, minsn=2,
                     zbest_include=True, fitarm=['b', 'r'], npoly=8)
    ts = F.open(single[0])['RVTAB'].data
    cfgf = str(tmp_path / 'c.yaml')
    with open(cfgf, 'w') as fp:
        yaml.safe_dump({k: v for k, v in CFG.items()
                        if k != 'config_file_path'}, fp)
    links = []
    for i in range(2):
        links.append(str(tmp_path / ('coadd-m%d.fits' % i)))
        os.symlink(COADD, links[-1])
        os.symlink(os.path.join(GOLD, 'redrock-golden.fits'),
                   str(tmp_path / ('redrock-m%d.fits' % i)))
    lst = str(tmp_path / 'files.txt')
    with open(lst, 'w') as fp:
        fp.write('\n'.join(links) + '\n')
    st = str(tmp_path / 'status')
    argv = ['--config', cfgf, '--input_file_from', lst, '--output_dir',
            str(tmp_path / 'out'), '--minsn', '2', '--zbest_include', '--no_subdirs',
            '--fitarm', 'B,R', '--npoly', '8', '--process_status_file', st,
            '--output_tab_prefix', 'tab', '--output_mod_prefix', 'mod',
            '--log_level', 'ERROR', '--files_per_batch', '2']
    D.main(argv)
    rows = [l.split() for l in open(st).read().strip().split('\n')]
    assert sorted(r[0] for r in rows) == sorted(links)
    assert all(r[1] == 'SUCCESS' and int(r[2]) == n1 for r in rows), rows
    for i in range(2):
        hd = F.open(str(tmp_path / 'out' / ('tab_coadd-m%d.fits' % i)))
        tg = hd['RVTAB'].data
        assert tg.columns.names == ts.columns.names
        for c in ts.columns.names:
            a, b = ts[c], tg[c]
            assert np.array_equal(a, b) if a.dtype.kind in 'SUb' else \
                np.array_equal(a, b, equal_nan=True), c
        assert hd[0].header['RVS_CMD'] == ' '.join(argv)
        assert os.path.exists(str(tmp_path / 'out' / ('mod_coadd-m%d.fits' % i)))
    # the same list as a shared queue file (--queue_file): emptied, same products
    with open(lst, 'w') as fp:
        fp.write('\n'.join(links) + '\n')
    st2 = str(tmp_path / 'status2')
    D.main(['--config', cfgf, '--input_file_from', lst, '--queue_file', '--output_dir',
            str(tmp_path / 'outq'), '--minsn', '2', '--zbest_include', '--no_subdirs',
            '--fitarm', 'B,R', '--npoly', '8', '--process_status_file', st2,
            '--log_level', 'ERROR', '--files_per_batch', '2'])
    assert open(lst).read() == ''
    rows = [l.split() for l in open(st2).read().strip().split('\n')]
    assert sorted(r[0] for r in rows) == sorted(links)
    assert all(r[1] == 'SUCCESS' and int(r[2]) == n1 for r in rows), rows
    tq = F.open(str(tmp_path / 'outq' / 'rvtab_coadd-m1.fits'))['RVTAB'].data
    assert np.array_equal(tq['VRAD'], ts['VRAD'], equal_nan=True)


def test_proc_many_two_fit_threads(dcases, desi_libs, tmp_path, monkeypatch):
    """RVS_DESI_FIT_THREADS=2: two groups of files fitted side by side by two threads
    -- every product is the one of the file processed alone, bit for bit, and every
    file has its status line"""
    import yaml
    from rvspecfit_amd import fits_min as F
    from rvspecfit_amd.desi import desi_fit as D
    kw = dict(doplot=False, minsn=2, zbest_include=True)
    single = (str(tmp_path / 's_tab.fits'), str(tmp_path / 's_mod.fits'))
    n1 = D.proc_desi(COADD, single[0], single[1], None, CFG, **kw)
    ts = F.open(single[0])['RVTAB'].data
    cfgf = str(tmp_path / 'c.yaml')
    with open(cfgf, 'w') as fp:
        yaml.safe_dump({k: v for k, v in CFG.items()
                        if k != 'config_file_path'}, fp)
    links = []
    for i in range(7):
        links.append(str(tmp_path / ('coadd-c%d.fits' % i)))
        os.symlink(COADD, links[-1])
        os.symlink(os.path.join(GOLD, 'redrock-golden.fits'),
                   str(tmp_path / ('redrock-c%d.fits' % i)))
    st = str(tmp_path / 'status')
    monkeypatch.setenv('RVS_DESI_FIT_THREADS', '2')
    D.proc_many(links, str(tmp_path / 'out'), 'rvtab', 'rvmod', config_fname=cfgf,
                minsn=2, zbest_include=True, doplot=False, subdirs=False,
                process_status_file=st, shard=(0, 1), files_per_batch=2)
    rows = [l.split() for l in open(st).read().strip().split('\n')]
    assert sorted(r[0] for r in rows) == sorted(links)
    assert all(r[1] == 'SUCCESS' and int(r[2]) == n1 for r in rows), rows
    for i in range(7):
        tg = F.open(str(tmp_path / 'out' / ('rvtab_coadd-c%d.fits' % i)))['RVTAB'].data
        for c in ts.columns.names:
            a, b = ts[c], tg[c]
            assert np.array_equal(a, b) if a.dtype.kind in 'SUb' else \
                np.array_equal(a, b, equal_nan=True), (i, c)


def test_proc_many_worker_processes(dcases, tmp_path):
    """nthreads = 2: two worker processes share the GPU, each with its stride
    of the file list (the library comes from disk: converted-artefact files in
    config['template_lib']); products identical to the single-process run"""
    import shutil
    import yaml
    from rvspecfit_amd import fits_min as F
    from rvspecfit_amd.desi import desi_fit as D
    tl = tmp_path / 'templ'
    tl.mkdir()
    for n in ('desi_b', 'desi_r', 'desi_z'):
        shutil.copy(os.path.join(GOLD, 'lib_%s.npz' % n),
                    str(tl / ('rvsgpu_%s.npz' % n)))
    cfgf = str(tmp_path / 'c.yaml')
    cfg = {k: v for k, v in CFG.items() if k != 'config_file_path'}
    cfg['template_lib'] = str(tl) + '/'
    with open(cfgf, 'w') as fp:
        yaml.safe_dump(cfg, fp)
    links = []
    for i in range(4):
        links.append(str(tmp_path / ('coadd-c%d.fits' % i)))
        os.symlink(COADD, links[-1])
        os.symlink(os.path.join(GOLD, 'redrock-golden.fits'),
                   str(tmp_path / ('redrock-c%d.fits' % i)))
    outs = {}
    for nthr in (1, 2):
        od = str(tmp_path / ('out%d' % nthr))
        st = str(tmp_path / ('status%d' % nthr))
        D.proc_many(links, od, 'rvtab', 'rvmod', config_fname=cfgf, minsn=2,
                    zbest_include=True, doplot=False, subdirs=False,
                    process_status_file=st, shard=(0, 1), files_per_batch=2,
                    nthreads=nthr)
        rows = sorted(l.split()[:3] for l in open(st).read().strip().split('\n'))
        assert [r[0] for r in rows] == sorted(links)
        assert all(r[1] == 'SUCCESS' and int(r[2]) == 10 for r in rows)
        outs[nthr] = [F.open(os.path.join(od, 'rvtab_coadd-c%d.fits' % i)
                             )['RVTAB'].data for i in range(4)]
    for a, b in zip(outs[1], outs[2]):
        for c in a.columns.names:
            x, y = a[c], b[c]
            assert np.array_equal(x, y) if x.dtype.kind in 'SUb' else \
                np.array_equal(x, y, equal_nan=True), c


# --------------------------------------------------------------------------
# the driver on MLP libraries (BASELINE configs[3] through desi_fit)
# --------------------------------------------------------------------------
MLP_ARMS = dict(b=(4000., 4600.), r=(6000., 6600.), z=(8000., 8600.))
MLP_CFG = dict(template_lib='mlp-desi://', min_vel=-1000, max_vel=1000,
               min_vel_step=0.2, vel_step0=5, min_vsini=0.1, max_vsini=500,
               second_minimizer=False, config_file_path='/x/config.yaml')


def _mlp_weights(ntp, seed):
    rng = np.random.RandomState(seed)
    dims = np.array([4, 48, 48, ntp], dtype=np.int32)
    d = dict(nn_dims=dims, nn_M=np.array([3.8, 2.5, -1., 0.5]),
             nn_S=np.array([0.17, 1.4, 0.6, 0.3]))
    for i in range(3):
        k, n = dims[i], dims[i + 1]
        d['nn_W%d' % i] = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
        d['nn_b%d' % i] = (0.05 * rng.standard_normal(n)).astype(np.float32)
    d['nn_W2'] *= 0.2   # exp(output) of order one, features of ~10 %
    return d


def _mlp_dicts():
    """three short DESI-like setups whose evaluator is a seeded MLP; the CCF set is
    built from the MLP's own templates at the nodes of a 4^4 grid"""
    from oracle import rvs_oracle as orc
    from rvspecfit_amd import synth
    out = {}
    for a, (l0, l1) in MLP_ARMS.items():
        lib = synth.make_interp_library_fast('desi_' + a, l0, l1, 0.4,
                                             grid_kw=dict(nteff=4, nlogg=4, nfeh=4,
                                                          nalpha=4))
        w = _mlp_weights(len(lib['lam']), 50 + ord(a))
        ws = [(w['nn_W%d' % i], w['nn_b%d' % i]) for i in range(3)]
        rows = orc.nn_forward(ws, lib['physical_vec'].T, w['nn_M'], w['nn_S'])
        lib['dats'] = np.log(rows).astype(np.float32)
        ccf = synth.make_ccf_templates(
            lib, l0, l1, 0.4, every=8, vsinis=(0., 300.),
            convolve=lambda lam, t, v: orc.convolve_vsini_rows(lam, t, v), cont=1.0)
        d = synth.library_as_npz_dict(lib, ccf)
        for k in ('dats', 'idgrid', 'vec', 'uvec0', 'uvec1', 'uvec2', 'uvec3'):
            d.pop(k)
        d.update(w)
        out['desi_' + a] = d
    return out


def test_proc_desi_on_mlp_library(tmp_path):
    """desi_fit.proc_desi on libraries whose evaluator is an MLP (nn/
    RVSInterpolator.py:36-42 as rvs_template_nn; the optimiser's rounds inside
    rvs_nm_run on rvs_template_nn_arms + rvs_objective_from_template): RVS_WARN,
    SUCCESS, VRAD and chi^2 of every fibre against the oracle's ccf_fit + process on
    the driver's own conditioned spectra and the warning rules of desi_fit.py:381-441.
    Spectra drawn from the library itself (its template at the truth point, at the
    truth velocity) + noise: ordinary fibres succeed; one fibre of pure noise does
    not (CHISQ_WARN).  Round 5's bench reported success_frac 0.0 on MLP libraries:
    its spectra came from another template family than its random-weight MLP."""
    import torch
    from oracle import rvs_oracle as orc
    from rvspecfit_amd import engine, fits_min as F, spec_fit, spec_inter
    from rvspecfit_amd.desi import desi_fit as D
    from rvspecfit_amd.library import TemplateLibrary
    dicts = _mlp_dicts()
    for n, d in dicts.items():
        spec_inter.register_library(TemplateLibrary(n, d), MLP_CFG['template_lib'])
    olibs = {n: orc.make_library(d) for n, d in dicts.items()}
    nf = 6
    rng = np.random.RandomState(77)
    truth = np.stack([rng.uniform(4500, 9000, nf), rng.uniform(1, 4, nf),
                      rng.uniform(-1.6, -0.4, nf), rng.uniform(0.2, 0.8, nf)], axis=1)
    tvel = rng.uniform(-200, 200, nf)
    snr = np.array([30., 100., 60., 25., 150., 40.])
    lams = {a: np.arange(l0 + 40, l1 - 40, 0.8) for a, (l0, l1) in MLP_ARMS.items()}
    one = [engine.ArmData('desi_' + a, lams[a], np.ones((nf, len(lams[a]))),
                          np.ones((nf, len(lams[a])))) for a in MLP_ARMS]
    raw = spec_fit.get_chisq(engine.SpecBatch(one), torch.as_tensor(tvel).to('cuda'),
                             torch.as_tensor(truth).to('cuda'), None, None,
                             options=dict(npoly=10), config=MLP_CFG,
                             full_output=True)['raw_models']
    fname = str(tmp_path / 'coadd-mlp.fits')
    hdus = [F.PrimaryHDU()]
    hdus[0].header['SPGRP'] = 'healpix'
    fm = F.FitsTable()
    fm.add('TARGETID', np.arange(nf, dtype=np.int64) + 39628000000000000)
    fm.add('FIBER', np.arange(nf, dtype=np.int32))
    fm.add('TARGET_RA', np.linspace(150., 151., nf))
    fm.add('TARGET_DEC', np.linspace(2., 3., nf))
    fm.add('OBJTYPE', np.array(['TGT'] * nf))
    fm.add('COADD_FIBERSTATUS', np.zeros(nf, dtype=np.int32))
    fm.add('BRICKID', np.zeros(nf, dtype=np.int32))
    hdus.append(F.BinTableHDU(fm, name='FIBERMAP'))
    sc = F.FitsTable()
    sc.add('TARGETID', fm['TARGETID'])
    for a, r in zip(MLP_ARMS, raw):
        sp0 = r.cpu().numpy() * (1 + 0.1 * (lams[a] - lams[a].mean()) / 500.)
        es = sp0 / snr[:, None]
        spec = sp0 + es * rng.normal(size=sp0.shape)
        spec[nf - 1] = 1.0 + es[nf - 1] * rng.normal(size=sp0.shape[1])  # no star
        mask = (rng.uniform(size=sp0.shape) < 0.03).astype(np.int32)
        flux, ivar = spec.astype(np.float32), (1.0 / es**2).astype(np.float32)
        A = a.upper()
        hdus += [F.ImageHDU(lams[a], name=A + '_WAVELENGTH'),
                 F.ImageHDU(flux, name=A + '_FLUX'),
                 F.ImageHDU(ivar, name=A + '_IVAR'),
                 F.ImageHDU(mask, name=A + '_MASK')]
        sc.add('MEDIAN_COADD_SNR_' + A, D.get_sns(flux, ivar, mask).astype(np.float64))
    hdus.append(F.BinTableHDU(sc, name='SCORES'))
    F.HDUList(hdus).writeto(fname)
    tabf, modf = str(tmp_path / 'rvtab.fits'), str(tmp_path / 'rvmod.fits')
    n = D.proc_desi(fname, tabf, modf, None, MLP_CFG, doplot=False, minsn=-1e9,
                    npoly=10)
    assert n == nf
    tab = F.open(tabf)['RVTAB'].data
    # ---- the oracle on the driver's own conditioned spectra
    FP = F.open(fname)
    setups = list(MLP_ARMS)
    fluxes, ivars, masks, waves, resolutions = D.read_data(FP, setups)
    opt = dict(npoly=10)
    names = ['teff', 'logg', 'feh', 'alpha']
    want = dict(vel=[], chisq=[], chisq_c=[], vsini=[], vel_err=[], bad=[],
                teff=[], feh=[], logg=[])
    for i in range(nf):
        sds0 = D.get_specdata(waves, fluxes, ivars, masks, resolutions, i, setups)
        sds = [orc.SpecData(s.name, s.lam, s.spec, s.espec, badmask=s.badmask)
               for s in sds0]
        c = orc.ccf_fit(sds, MLP_CFG, olibs)
        pd0 = dict(zip(names, c['best_par']))
        if np.isfinite(c['best_vsini']):
            pd0['vsini'] = float(c['best_vsini'])
        r = orc.process(sds, pd0, None, opt, MLP_CFG, olibs)
        cc = orc.get_chisq_continuum(sds, options=opt)
        want['vel'].append(r['vel'])
        want['vel_err'].append(r['vel_err'])
        want['chisq'].append(np.sum(r['chisq_array']))
        want['chisq_c'].append(np.sum(cc['chisq_array']))
        want['vsini'].append(r['vsini'] if r['vsini'] is not None else np.nan)
        want['bad'].append(r['bad_hessian'])
        for k in ('teff', 'feh', 'logg'):
            want[k].append(r['param'][k])
    w_ref = D.rvs_warn_bits(want['chisq'], want['chisq_c'], want['vel'],
                            want['vsini'], want['vel_err'], want['bad'],
                            want['teff'], want['feh'], want['logg'], MLP_CFG)
    chi, chi_ref = tab['CHISQ_TOT'], np.array(want['chisq'])
    # (Nelder-Mead's fatol is 1e-3: an end point this far above or below the
    # oracle's is the same fit)
    same = np.abs(chi - chi_ref) <= 5e-3
    assert same.sum() >= nf - 1, (chi, chi_ref)
    assert np.all(chi <= chi_ref + 0.05)
    np.testing.assert_allclose(tab['CHISQ_C_TOT'], want['chisq_c'], rtol=1e-8)
    assert np.abs(tab['VRAD'] - np.array(want['vel']))[same].max() <= 0.01
    bh = D.bitmasks['BAD_HESSIAN']
    w = tab['RVS_WARN']
    assert np.array_equal(w & ~bh, w_ref & ~bh), (w, w_ref)
    assert np.array_equal(w[same], w_ref[same])
    assert np.array_equal(tab['SUCCESS'], w == 0)
    # the star-less fibre carries the chi^2 warning, the others are fits that mean
    # something: recovered velocities, no warning
    assert w[nf - 1] & D.bitmasks['CHISQ_WARN']
    assert (w[:nf - 1] == 0).sum() >= nf - 2
    ok = w[:nf - 1] == 0
    assert np.abs(tab['VRAD'][:nf - 1] - tvel[:nf - 1])[ok].max() < 5 * max(
        1.0, tab['VRAD_ERR'][:nf - 1][ok].max())
