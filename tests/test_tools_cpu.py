"""Host-side pieces (no GPU): CCF tables, the artefact converter, synth."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import scipy.interpolate

from rvspecfit_amd import ccf_tables as ct
from rvspecfit_amd import synth
from oracle import rvs_oracle as orc

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_interp_spline_tables_match_scipy():
    rng = np.random.RandomState(3)
    lam = np.arange(3600., 5800.1, 0.8)
    nodes, edges = ct.continuum_nodes(lam, 7934.)
    Eb, El, Cinv, istart = ct.interp_spline_tables(nodes, lam)
    p = rng.standard_normal(len(nodes))
    c = Cinv @ p
    mine = np.array([Eb[k] @ c[El[k]:El[k] + 3] for k in range(len(lam))])
    ref = scipy.interpolate.UnivariateSpline(nodes, p, s=0, k=2)(lam)
    np.testing.assert_allclose(mine, ref, rtol=1e-12, atol=1e-13)
    assert istart[0] == 0 and istart[-1] == len(lam)
    assert np.all(np.diff(istart) >= 0)
    np.testing.assert_allclose(ct.interp_spline_design(nodes, lam) @ p, ref,
                               rtol=1e-12, atol=1e-13)


def test_bin_ranges_match_binned_statistic():
    import scipy.stats
    lam = np.arange(4400., 4720.1, 0.8)
    nodes, edges = ct.continuum_nodes(lam, 1055.)
    rng = np.random.RandomState(4)
    y = rng.standard_normal(len(lam))
    bs = scipy.stats.binned_statistic(lam, y, 'median', bins=edges).statistic
    br = ct.bin_ranges(lam, edges)
    for j in range(len(edges) - 1):
        seg = y[br[j]:br[j + 1]]
        if len(seg):
            assert np.median(seg) == bs[j]
        else:
            assert np.isnan(bs[j])


def test_lag_and_rebin_tables_match_oracle(gold_libs):
    cc = gold_libs['gold_b'].ccf
    s1, i1, v1 = ct.lag_tables(cc['logl0'], cc['logl1'], cc['npoints'], 1000)
    s2, i2, v2 = orc.ccf_lag_tables(cc['logl0'], cc['logl1'], cc['npoints'], 1000)
    assert s1 == s2
    np.testing.assert_array_equal(i1, i2)
    np.testing.assert_array_equal(v1, v2)


def test_synth_batch_equals_scalar():
    lam = np.arange(4400, 4720.1, 0.8)
    a = synth.spectrum(lam, 5100., 2.2, -0.7, 0.15, wresol=0.3)
    b = synth.spectra_batch(lam, np.array([5100.]), np.array([2.2]),
                            np.array([-0.7]), np.array([0.15]), wresol=0.3)[0]
    np.testing.assert_allclose(a, b, rtol=1e-14)


@pytest.mark.skipif(not (os.path.exists('/opt/conda/bin/python3.9')
                         and os.path.exists('/tmp/golden_work/templ/interp_gold_b.h5')),
                    reason='needs the build container (h5py interpreter + the '
                           'reference artefacts made by make_golden.py)')
def test_convert_artefacts_reproduces_reference_loader(tmp_path):
    src = '/tmp/golden_work/templ'
    for f in os.listdir(src):
        if 'gold_b' in f and not f.startswith('rvsgpu_'):
            shutil.copy(os.path.join(src, f), tmp_path)
    subprocess.check_call(['/opt/conda/bin/python3.9', '-W', 'ignore',
                           os.path.join(REPO, 'tools', 'convert_artefacts.py'),
                           str(tmp_path), 'gold_b'], stdout=subprocess.DEVNULL)
    a = np.load(os.path.join(tmp_path, 'rvsgpu_gold_b.npz'))
    fixtures = [np.load(os.path.join(REPO, 'tests', 'golden', 'lib_gold_b.npz'))]
    if os.path.exists(os.path.join(src, 'ccf_nocont_gold_b.h5')):
        # the --nocontinuum set (make_golden_nocont.py) lands under ccfnc_*
        fixtures.append(np.load(os.path.join(REPO, 'tests', 'golden',
                                             'lib_nocont_gold_b.npz')))
        assert 'ccfnc_fft' in a.files
    for b in fixtures:
        for k in b.files:
            x, y = a[k], b[k]
            assert x.shape == y.shape, k
            if x.dtype.kind in 'USib':
                assert np.array_equal(x, y), k
            else:
                np.testing.assert_array_equal(x, y, err_msg=k)


def test_lockstep_neldermead_equals_scipy():
    """tests/refmachines/neldermead_torch follows scipy's Nelder-Mead (the optimiser of
    vel_fit.py:627-637) iteration for iteration: same nit, nfev, final simplex"""
    import scipy.optimize as so
    import torch
    from refmachines import neldermead_torch as nm
    rng = np.random.RandomState(1)
    S, N = 24, 6
    A = rng.normal(size=(S, N, N))
    A = np.einsum('sij,skj->sik', A, A) + np.eye(N)
    c = rng.normal(size=(S, N))

    def f1(i, x):
        d = x - c[i]
        return d @ A[i] @ d + 0.3 * np.sum(np.abs(d)**1.5) + \
            (1e30 if x[0] > 5 else 0)

    def fb(idx, X):
        return torch.as_tensor(np.array(
            [f1(int(i), x) for i, x in zip(idx.numpy(), X.numpy())]))

    simp = rng.normal(size=(S, N + 1, N)) * 2
    for maxiter in (10000, 30):
        r = nm.minimize(fb, torch.as_tensor(simp), maxiter=maxiter)
        for i in range(S):
            q = so.minimize(lambda x: f1(i, x), simp[i, 0], method='Nelder-Mead',
                            options=dict(fatol=1e-3, xatol=1e-2,
                                         initial_simplex=simp[i],
                                         maxiter=maxiter, maxfev=np.inf))
            assert q.nit == int(r['nit'][i]) and q.nfev == int(r['nfev'][i])
            assert q.success == bool(r['success'][i])
            np.testing.assert_array_equal(q.final_simplex[0],
                                          r['final_simplex'][0][i].numpy())
            np.testing.assert_array_equal(q.x, r['x'][i].numpy())


def test_lockstep_bfgs_equals_scipy():
    """tests/refmachines/bfgs_scipy_restated restates scipy's BFGS (vel_fit.py:653-658) with its
    Wolfe line searches and 2-point gradient: identical nit, nfev, status and
    iterates, on a smooth objective and on one with 1e-9 of deterministic noise
    (the regime of the real objective: precision-loss exit through
    line_search_wolfe2)"""
    import warnings
    import scipy.optimize as so
    from refmachines import bfgs_scipy_restated as bfgs_ref
    from rvspecfit_amd import bfgs
    rng = np.random.RandomState(2)
    S, N = 12, 5
    A = rng.normal(size=(S, N, N))
    A = np.einsum('sij,skj->sik', A, A) + np.eye(N)
    c = rng.normal(size=(S, N))

    def f1(i, x, nz):
        d = x - c[i]
        v = 0.5 * d @ A[i] @ d + 0.1 * np.sum(d**4) + np.sum(np.cos(d))
        return v + nz * np.sin(1e9 * np.sum(x))

    H0 = np.diag(rng.uniform(0.5, 2, N))
    x0 = rng.normal(size=(S, N)) * 2
    for nz, want_status in ((0.0, 0), (1e-9, 2)):
        r = bfgs_ref.minimize_lockstep(
            lambda idx, X: np.array([f1(int(i), x, nz) for i, x in zip(idx, X)]),
            x0, hess_inv0=H0, max_rows=17)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            for i in range(S):
                q = so.minimize(lambda x: f1(i, x, nz), x0[i], method='BFGS',
                                options=dict(hess_inv0=H0))
                assert q.nit == r['nit'][i] and q.nfev == r['nfev'][i]
                assert q.status == r['status'][i] == want_status
                np.testing.assert_array_equal(q.x, r['x'][i])
                assert q.fun == r['fun'][i]


def test_native_bfgs_follows_scipy_restatement():
    """The C++ lock-step BFGS (csrc/bfgs_host.cpp, rvs_bfgs_*) against the
    scipy-pinned Python restatement above.  Smooth objectives: the same nit,
    nfev and status for every run, the same function value to 1e-9 (the scalar
    C++ sums are not numpy's BLAS sums, and the forward-difference gradient
    amplifies the last bit by 1/1.5e-8: along flat directions the optimum
    itself moves by up to 1e-3).  A Rosenbrock-type valley exercises the
    line_search_wolfe2 / zoom fall-back and long runs; 1e-9 of noise the
    precision-loss exit, where only the statistics are comparable."""
    from refmachines import bfgs_scipy_restated as bfgs_ref
    from rvspecfit_amd import bfgs
    rng = np.random.RandomState(5)
    S, N = 40, 6
    A = rng.normal(size=(S, N, N))
    A = np.einsum('sij,skj->sik', A, A) + np.eye(N)
    c = rng.normal(size=(S, N))

    def quartic(idx, X, nz=0.0):
        d = X - c[idx]
        v = 0.5 * np.einsum('ji,jik,jk->j', d, A[idx], d) + \
            0.1 * np.sum(d**4, axis=1) + np.sum(np.cos(d), axis=1)
        return v + nz * np.sin(1e9 * np.sum(X, axis=1))

    def valley(idx, X):
        return np.sum(100.0 * (X[:, 1:] - X[:, :-1]**2)**2 +
                      (1 - X[:, :-1])**2, axis=1)

    H0 = np.diag(rng.uniform(0.5, 2, N))
    x0 = rng.normal(size=(S, N)) * 2
    a = bfgs_ref.minimize_lockstep(quartic, x0, hess_inv0=H0, max_rows=97)
    b = bfgs.minimize_lockstep_native(quartic, x0, hess_inv0=H0, max_rows=97)
    assert np.array_equal(a['status'], b['status']) and not a['status'].any()
    assert np.array_equal(a['nit'], b['nit'])
    assert np.array_equal(a['nfev'], b['nfev'])
    assert np.abs(a['x'] - b['x']).max() < 5e-3
    assert np.allclose(a['fun'], b['fun'], rtol=1e-9, atol=1e-9)
    assert a['nit'].max() > 8
    # the valley: ~100 iterations per run, wolfe2 / zoom fall-backs, and an exit
    # (converged or precision loss) that depends on the last bits -- the runs
    # end in the same minimum with similar effort
    a = bfgs_ref.minimize_lockstep(valley, x0, max_rows=97)
    b = bfgs.minimize_lockstep_native(valley, x0, max_rows=97)
    assert set(a['status']) <= {0, 2} and set(b['status']) <= {0, 2}
    assert np.allclose(a['fun'], b['fun'], atol=1e-6)
    assert np.abs(a['x'] - b['x']).max() < 5e-3
    assert b['nit'].max() > 30
    assert abs(a['nfev'].mean() - b['nfev'].mean()) < 0.1 * a['nfev'].mean()
    noisy = lambda idx, X: quartic(idx, X, 1e-9)  # noqa: E731
    a = bfgs_ref.minimize_lockstep(noisy, x0, hess_inv0=H0)
    b = bfgs.minimize_lockstep_native(noisy, x0, hess_inv0=H0)
    assert (a['status'] == 2).mean() > 0.8 and (b['status'] == 2).mean() > 0.8
    # (gradient noise 1e-9 / 1.5e-8: both stop a few 1e-3 above the minimum)
    assert abs(a['fun'].mean() - b['fun'].mean()) < 0.02
    assert abs(a['nfev'].mean() - b['nfev'].mean()) < 0.25 * a['nfev'].mean()
    # argument checks
    with pytest.raises(ValueError):
        bfgs.minimize_lockstep_native(quartic, np.zeros((2, 17)))


def test_native_bfgs_under_sanitizers(tmp_path):
    """csrc/bfgs_host.cpp (C++20 coroutines, hand-managed frames) built with
    g++ -fsanitize=address,undefined and driven through 200 Rosenbrock runs,
    some of them with noise: no report, all runs end"""
    import shutil
    import subprocess
    if shutil.which('g++') is None:
        pytest.skip('no g++')
    exe = str(tmp_path / 'bfgs_san')
    cmd = ['g++', '-std=c++20', '-O1', '-g', '-fsanitize=address,undefined',
           '-fno-omit-frame-pointer', '-I' + os.path.join(REPO, 'include'), '-o',
           exe, os.path.join(REPO, 'tests', 'bfgs_sanitizer_main.cpp'),
           os.path.join(REPO, 'rvspecfit_amd', 'csrc', 'bfgs_host.cpp')]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'rounds' in out.stdout and 'ERROR' not in out.stderr
    assert 'runtime error' not in out.stderr
