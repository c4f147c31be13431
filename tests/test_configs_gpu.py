"""BASELINE.json configs[1] and configs[3] at their full sizes, each with a sample
of the batch compared with the CPU oracle (north-star tolerances: integer work
exact, RV 0.01 km/s, chi^2 1e-6 relative) and the size-independent invariance
properties of tests/test_full_size.py.

  configs[1]  1 000 synthetic spectra, 1 arm linspace(4000, 5000, 2001), template
              grid 3950-5050 A step 0.5 (2272 px), N_fft 4096, polylinear + CCF
  configs[3]  DESI b/r/z, 10 000 spectra, the NN (MLP on MFMA) template evaluator
"""
import argparse
import contextlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def bench_setup(arms=None, evaluator=None):
    """bench.py keeps the workload in module globals (it is a script)"""
    import bench
    keep = bench.ARMS, bench.EVALUATOR
    if arms is not None:
        bench.ARMS = arms
    if evaluator is not None:
        bench.EVALUATOR = evaluator
    try:
        yield bench
    finally:
        bench.ARMS, bench.EVALUATOR = keep


def _gpu_convolve(dev):
    from rvspecfit_amd import engine

    def f(lam, templ, vsini):
        t = torch.as_tensor(np.ascontiguousarray(templ)).to(dev)
        v = torch.as_tensor(np.ascontiguousarray(vsini)).to(dev)
        return engine.convolve_vsini(lam, t, v).cpu().numpy()
    return f


def _compare_with_oracle(bench, arms, rec, ix, workload, evaluator, npix_tot,
                         chi_rtol):
    from rvspecfit_amd import pipeline
    F = pipeline.RECORD_FIELDS
    dev = rec.device
    sub = [(nm, lam, sp[ix.to(dev)], es[ix.to(dev)], bad[ix.to(dev)])
           for nm, lam, sp, es, bad in arms]
    args = argparse.Namespace(ccf_every=64, cpu_cores=8, workload=workload,
                              evaluator=evaluator)
    cb = bench.run_cpu_baseline(sub, len(ix), args)
    o = np.array(cb['recs'])
    g = rec[ix.to(dev)].cpu().numpy()
    assert np.array_equal(g[:, F.index('best_id')], o[:, 0])        # index work
    assert np.abs(g[:, F.index('vrad_ccf')] - o[:, 1]).max() < 1e-2   # km/s
    assert np.abs(g[:, F.index('best_vel')] - o[:, 2]).max() < 1e-2
    np.testing.assert_allclose(g[:, F.index('vel_err')], o[:, 3], rtol=1e-3)
    # -2 log L passes through zero: relative to max(|chi|, pixel count)
    rel = np.abs(g[:, F.index('best_chi')] - o[:, 4]) / \
        np.maximum(np.abs(o[:, 4]), npix_tot)
    assert rel.max() < chi_rtol, rel
    na = len(arms)
    for ia in range(na):   # continuum-only chi^2 per arm
        np.testing.assert_allclose(g[:, F.index('chisq_c%d' % ia)], o[:, 5 + ia],
                                   rtol=1e-8)
    return g, o


def test_config0_one_spectrum_through_the_reference_api():
    """BASELINE configs[0] at its exact shape -- ONE synthetic spectrum, one arm
    linspace(4000, 5000, 2001), template grid 3950-5050 A (2272 px), N_fft 4096,
    polylinear -- through the reference's own single-spectrum API (a list of
    SpecData in, floats / dicts out, as tests/test_fit_fake.py drives it):
    fitter_ccf.fit, spec_fit.find_best on the 400-velocity grid, get_chisq,
    get_chisq_continuum and vel_fit.process against the oracle on the same
    spectrum (index work exact, RV 0.01 km/s, chi^2 1e-6)."""
    from oracle import rvs_oracle as orc
    from rvspecfit_amd import _lib, fitter_ccf, spec_fit, spec_inter, vel_fit
    from rvspecfit_amd.library import TemplateLibrary
    _lib.require_gpu()
    dev = torch.device('cuda', 0)
    with bench_setup(arms=('c', )) as bench:
        cfg = dict(bench.CONFIG, template_lib='synthetic://cfg0', max_vsini=500,
                   second_minimizer=False)
        dicts = bench.build_library_dicts(64, _gpu_convolve(dev))
        for name, dd in dicts.items():
            spec_inter.register_library(TemplateLibrary(name, dd, device=dev),
                                        cfg['template_lib'])
        olibs = {k: orc.make_library(v) for k, v in dicts.items()}
        tp = bench.truth_params(1, seed=5)
        name, lam, sp, es, bad = bench.make_spectra_device(tp, dev)[0]
        assert np.array_equal(lam, np.linspace(4000, 5000, 2001))
        sp, es, bad = (x[0].cpu().numpy() for x in (sp, es, bad))
        sds = [spec_fit.SpecData(name, lam, sp, es, badmask=bad != 0)]
        osd = [orc.SpecData(name, lam, sp, es, badmask=bad != 0)]
        opt = bench.OPTIONS
        # CCF
        r = fitter_ccf.fit(sds, cfg)
        o = orc.ccf_fit(osd, cfg, olibs)
        assert abs(r['best_vel'] - o['best_vel']) < 1e-2
        assert [r['best_par'][k] for k in ('teff', 'logg', 'feh', 'alpha')] == \
            list(o['best_par'])
        # the 400-velocity grid at the CCF parameters
        vg = np.arange(cfg['min_vel'], cfg['max_vel'], cfg['vel_step0'])
        vs = o['best_vsini']
        rot = None if (vs is None or np.isnan(vs)) else (float(vs), )
        fb = spec_fit.find_best(sds, vg, [tuple(o['best_par'])], rot, options=opt,
                                config=cfg)
        grid = orc.chisq_grid_fast(osd, vg, o['best_par'], rot, opt, cfg, olibs)
        s = orc.grid_summary(vg, grid[:, None])
        assert abs(fb['best_vel'] - s['best_vel']) < 1e-2
        assert abs(fb['best_chi'] - s['best_chi']) <= 1e-6 * max(abs(s['best_chi']),
                                                                 2001)
        assert abs(fb['vel_err'] / s['vel_err'] - 1) < 1e-3
        # one point, and the continuum-only fit
        val = spec_fit.get_chisq(sds, float(s['best_vel']), tuple(o['best_par']),
                                 rot, options=opt, config=cfg)
        want = orc.get_chisq(osd, float(s['best_vel']), tuple(o['best_par']), rot,
                             options=opt, config=cfg, libs=olibs)
        assert abs(val - want) <= 1e-6 * max(abs(want), 2001)
        np.testing.assert_allclose(
            spec_fit.get_chisq_continuum(sds, options=opt)['chisq_array'],
            orc.get_chisq_continuum(osd, options=opt)['chisq_array'], rtol=1e-8)
        # vel_fit.process from the CCF point
        pd0 = dict(zip(('teff', 'logg', 'feh', 'alpha'), o['best_par']))
        if rot is not None:
            pd0['vsini'] = rot[0]
        p = vel_fit.process(sds, dict(pd0), options=opt, config=cfg)
        q = orc.process(osd, dict(pd0), None, opt, cfg, olibs)
        assert isinstance(p['vel'], float) and isinstance(p['param'], dict)
        assert abs(p['chisq'] - q['chisq']) < 5e-3          # fatol-level
        assert abs(p['vel'] - q['vel']) < max(0.01, 0.02 * q['vel_err'])
        assert abs(p['vel'] - tp['vel'][0]) < max(10, 3 * p['vel_err'])


def test_config1_1000_spectra_one_arm():
    """BASELINE configs[1] exactly"""
    from rvspecfit_amd import _lib, engine, pipeline, spec_inter
    from rvspecfit_amd.library import TemplateLibrary
    _lib.require_gpu()
    dev = torch.device('cuda', 0)
    S = 1000
    with bench_setup(arms=('c', )) as bench:
        cfg = dict(bench.CONFIG, template_lib='synthetic://cfg2')
        dicts = bench.build_library_dicts(64, _gpu_convolve(dev))
        d = dicts['desi_c']
        assert len(d['lam']) == 2272 and int(d['ccf_npoints']) == 4096
        for name, dd in dicts.items():
            spec_inter.register_library(TemplateLibrary(name, dd, device=dev),
                                        cfg['template_lib'])
        arms = bench.make_spectra_device(bench.truth_params(S, seed=2), dev)
        assert len(arms) == 1
        assert np.array_equal(arms[0][1], np.linspace(4000, 5000, 2001))
        batch = engine.SpecBatch([engine.ArmData(n, lam, sp, es, bad, device=dev)
                                  for n, lam, sp, es, bad in arms])
        # the reference's config keys are read from `cfg`; the oracle leg reads
        # bench.CONFIG (same values, other library name)
        fit = lambda b: pipeline.fit_batch(b, cfg, options=bench.OPTIONS)
        rec = fit(batch)
        F = pipeline.RECORD_FIELDS
        assert rec.shape == (S, pipeline.NREC)
        r = rec.cpu().numpy()
        assert np.isfinite(r[:, F.index('best_chi')]).all()
        tp = bench.truth_params(S, seed=2)
        assert np.median(np.abs(r[:, F.index('best_vel')] - tp['vel'])) < 5.0
        # a 16-spectrum sample against the oracle
        ix = torch.as_tensor([0, 1, 63, 64, 127, 200, 255, 256, 400, 511, 512,
                              640, 777, 900, 998, 999])
        _compare_with_oracle(bench, arms, rec, ix, 'cfg2', 'polylinear', 2001,
                             chi_rtol=1e-9)
        # position in the batch / batch size do not matter (bit for bit)
        g = torch.Generator(device='cpu')
        g.manual_seed(4)
        perm = torch.randperm(S, generator=g).to(dev)
        assert torch.equal(fit(batch.subset(perm)), rec[perm])
        sub = torch.arange(300, 365, device=dev)
        assert torch.equal(fit(batch.subset(sub)), rec[sub])


def test_config3_nn_evaluator_10000_spectra():
    """BASELINE configs[3]: the whole pipeline with the MLP evaluator at 10 000
    spectra; 8 of them against the oracle, whose evaluator is the numpy float32
    restatement of NNInterpolator.forward (oracle NNLibrary)"""
    from rvspecfit_amd import _lib, engine, pipeline, spec_inter
    from rvspecfit_amd.library import TemplateLibrary
    _lib.require_gpu()
    dev = torch.device('cuda', 0)
    S = 10000
    with bench_setup(evaluator='nn') as bench:
        cfg = dict(bench.CONFIG, template_lib='synthetic://desi_nn_full')
        dicts = bench.build_library_dicts(64, _gpu_convolve(dev))
        assert all('nn_dims' in d and 'dats' not in d for d in dicts.values())
        for name, dd in dicts.items():
            spec_inter.register_library(TemplateLibrary(name, dd, device=dev),
                                        cfg['template_lib'])
        arms = bench.make_spectra_device(bench.truth_params(S, seed=7), dev)
        batch = engine.SpecBatch([engine.ArmData(n, lam, sp, es, bad, device=dev)
                                  for n, lam, sp, es, bad in arms])
        try:
            rec = pipeline.fit_batch(batch, cfg, options=bench.OPTIONS)
            assert rec.shape == (S, pipeline.NREC)
            F = pipeline.RECORD_FIELDS
            assert torch.isfinite(rec[:, F.index('best_chi')]).all()
            ix = torch.as_tensor([0, 1234, 2500, 4999, 5000, 7321, 8888, 9999])
            npix_tot = sum(a[2].shape[1] for a in arms)
            # (float32 MLP: the template itself agrees to 3e-6 with the numpy
            # float32 oracle, MFMA summation order)
            _compare_with_oracle(bench, arms, rec, ix, 'desi', 'nn', npix_tot,
                                 chi_rtol=1e-6)
            # a subset of the batch, alone: bit for bit
            sub = torch.arange(4000, 4300, device=dev)
            assert torch.equal(pipeline.fit_batch(batch.subset(sub), cfg,
                                                  options=bench.OPTIONS),
                               rec[sub])
        finally:
            pass
