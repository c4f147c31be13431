#!/usr/bin/env python3
"""Headline benchmark: DESI 3-arm spectra/s for the full CCF + chi^2-grid fit.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the contract unit of work (SURVEY 8(d) D1:
fitter_ccf.fit -> template build at the CCF parameters -> find_best over 400
velocities -> get_chisq_continuum) over a batch of `--spectra` synthetic DESI
b/r/z spectra that are already resident in HBM.  Workload = BASELINE.json
configs[2] (DESI 3-arm, 10 000 spectra, polylinear interpolator, T=76 CCF
templates); with N>1 every rank holds its own 10 000-spectra shard (weak
scaling) and the only collective is the all_gather of the fixed-size result
records.

Rank 0 prints ONE JSON line with the contract keys plus
  roofline     algorithmic HBM bytes of the dominant kernel / its measured time
  cpu_baseline the oracle (CPU port of the reference) timed on host cores
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from rvspecfit_amd import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
GRID_KW = dict(nteff=7, nlogg=7, nfeh=7, nalpha=7)
RESOL = 3000.
ARMS = ('b', 'r', 'z')
# BASELINE configs[1] ("Cfg2"): 1 arm 4000-5000 A, 2001 px, template 3950-5050 A
# step 0.5 (2272 px), N_fft 4096
CFG2_ARM = dict(obs=(4000., 5000.01, 0.5), templ=(3950., 5050., 0.5))
# --workload sdss: ONE arm on the log-lambda lattice of an SDSS spectrum
# (log10 lam = 3.5798 + 1e-4 k, 3842 pixels, 3800-9200 A: the reference's own data
# fixture tests/data/spec-0266-51602-0031.fits), every object on its OWN piece of
# the lattice -- shifted start, truncated end (spec_fit.py:70-145 takes any `lam`
# per object; tests/test_sdss.py) -- i.e. a grid set of S grids
SDSS_ARM = dict(obs=(3.5798, 1e-4, 3842), templ=(3750., 9300., 1.0))
SDSS_PIECES = None   # int64 [S, 2] (first pixel, pixel count) of every spectrum
CONFIG = dict(min_vel=-1000, max_vel=1000, vel_step0=5, min_vel_step=0.2,
              min_vsini=0.1, max_vsini=500, template_lib='synthetic://desi')
OPTIONS = dict(npoly=10)
EVALUATOR = 'polylinear'


def arm_name(a):
    return 'desi_' + a


def arm_def(a):
    if a == 's':
        return SDSS_ARM
    return CFG2_ARM if a == 'c' else synth.DESI_ARMS[a]


def obs_lam(a):
    lo, hi, st = arm_def(a)['obs']
    if a == 's':   # (log10 of the first pixel, step in dex, pixels)
        return 10**(lo + hi * np.arange(st))
    return np.arange(lo, hi, st)


def sdss_pieces(S, seed):
    """(first pixel, pixel count) of every spectrum on the 3842-pixel lattice"""
    rng = np.random.RandomState(4242 + seed)
    a0 = rng.randint(0, 256, S)
    n = 3842 - a0 - rng.randint(0, 400, S)
    return np.stack([a0, n], axis=1).astype(np.int64)


def nn_weights(ntp, seed):
    """Seeded float32 MLP 4 -> 256 -> 256 -> 256 -> 200 -> ntp (the reference's
    default architecture, nn/train_interpolator.py:105-114).  There is no network
    to fetch trained checkpoints and training is out of scope, so the weights
    are random: the workload shape (GEMM sizes, bytes) is exact, the templates
    are not physical."""
    rng = np.random.RandomState(seed)
    dims = np.array([4, 256, 256, 256, 200, ntp], dtype=np.int32)
    d = dict(nn_dims=dims, nn_M=np.array([3.8, 2.5, -1., 0.5]),
             nn_S=np.array([0.17, 1.4, 0.6, 0.3]))
    for i in range(5):
        k, n = dims[i], dims[i + 1]
        d['nn_W%d' % i] = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
        d['nn_b%d' % i] = (0.05 * rng.standard_normal(n)).astype(np.float32)
    d['nn_W4'] *= 0.1  # keep exp(output) of order one
    return d


def nn_node_rows(w, physical_vec, chunk=4096):
    """log-flux rows (float32, what GridInterp keeps as `dats`) of the MLP `w` at
    the grid nodes physical_vec [4, n]: nn/NNInterpolator.py:14-91 with the
    Mapper's log10 of teff (:159-171) in float32 numpy -- the CCF template set of an
    MLP library is built from the MLP's own templates, as make_ccf.py builds it
    from whatever evaluator the setup has"""
    dims = [int(_) for _ in w['nn_dims']]
    M, Sd = w['nn_M'], w['nn_S']
    out = np.empty((physical_vec.shape[1], dims[-1]), dtype=np.float32)
    for a in range(0, out.shape[0], chunk):
        p = physical_vec[:, a:a + chunk].T.astype(np.float32)
        y = p.copy()
        y[:, 0] = np.log10(p[:, 0])
        h = ((y - M) / Sd).astype(np.float32)
        for i in range(len(dims) - 1):
            h = (h @ w['nn_W%d' % i].T + w['nn_b%d' % i]).astype(np.float32)
            if i < len(dims) - 2:
                h = (h / (np.float32(1) + np.exp(-h))).astype(np.float32)
        out[a:a + chunk] = h
    return out


_TRI_CACHE = {}


def node_templates(T, S):
    """pipeline.fit_batch's rule for sharing one template per CCF node"""
    ntp = max(len(synth.template_lam_grid(*arm_def(a)['templ'])) for a in ARMS)
    return T <= S and T * ntp * 32 <= (64 << 20)


def traffic_key(args, S, T, nfft, grid_name):
    """what a PMC traffic figure is valid for: the workload a line describes"""
    key = '%s|S=%d|T=%d|nfft=%d|%s|grid=%s|refine=%d|resol=%d|templ=%s' % (
        args.workload, S, T, nfft, args.evaluator, grid_name, int(args.refine),
        int(args.resolution_matrix),
        'spectrum' if (getattr(args, 'per_spectrum_templates', False)
                       or not node_templates(T, S)) else 'node')
    return key if OPTIONS['npoly'] == 10 else key + '|npoly=%d' % OPTIONS['npoly']


def ccf_every_for(ccf_every, ngrid):
    """--ccf-every is quoted for the 7^4 grid (64 -> T = 76, 9 -> T = 534); a
    grid of another size keeps the same NUMBER of CCF templates"""
    n0 = 7**4
    nsel = -(-n0 // ccf_every)
    return max(1, -(-ngrid // nsel)) if ngrid != n0 else ccf_every


def build_library_dicts(ccf_every, convolve, device=None):
    """Synthetic DESI-shape template libraries (7^4 grid unless --grid,
    6215/5303/6449 px, N_fft 8192) in the converted-artefact dict layout.
    `device`: synthesise the grid rows there (float32 device tensor)."""
    out = {}
    for a in ARMS:
        l0, l1, st = arm_def(a)['templ']
        lib = synth.make_interp_library_fast(arm_name(a), l0, l1, st,
                                             grid_kw=GRID_KW, resol=RESOL,
                                             device=device)
        every = ccf_every_for(ccf_every, lib['dats'].shape[0])
        w = None
        if EVALUATOR == 'nn':
            # the MLP replaces the polylinear evaluator of the template build: its
            # own templates at the grid nodes feed the CCF set (and, in main(), the
            # observed spectra are drawn from it: make_spectra_from_library), so
            # that the fits of this configuration mean something although the
            # weights are random
            w = nn_weights(len(lib['lam']), seed=11 + ord(a))
            lib['dats'] = nn_node_rows(w, lib['physical_vec'])
        ccf = synth.make_ccf_templates(lib, l0, l1, st, every=every,
                                       vsinis=(0., 300.), convolve=convolve,
                                       cont=None if w is None else 1.0)
        out[arm_name(a)] = synth.library_as_npz_dict(lib, ccf)
        if w is not None:
            d = out[arm_name(a)]
            for k in ('dats', 'idgrid', 'vec', 'uvec0', 'uvec1', 'uvec2', 'uvec3'):
                d.pop(k)
            d.update(w)
        if EVALUATOR == 'tri':
            # interpolation_type 'triangulation' (make_nd without --regulargrid,
            # spec_inter.py:11-59): scipy's Delaunay of the mapped grid nodes, its
            # arrays as the artefact converter exports them
            import scipy.spatial
            d = out[arm_name(a)]
            # (the arms share the parameter grid: one triangulation for all of them)
            key = lib['vec'].tobytes()
            if _TRI_CACHE.get('key') != key:
                _TRI_CACHE.update(key=key, dl=scipy.spatial.Delaunay(lib['vec'].T))
            dl = _TRI_CACHE['dl']
            for k in ('idgrid', 'uvec0', 'uvec1', 'uvec2', 'uvec3'):
                d.pop(k)
            rows = lib['dats']
            if not isinstance(rows, np.ndarray):
                rows = rows.cpu().numpy()
            d.update(dats=rows.astype(np.float64),
                     simplices=dl.simplices.astype(np.int32),
                     transform=dl.transform, extraflags=np.zeros(rows.shape[0]),
                     interpolation_type=np.array('triangulation'))
    return out


def truth_params(S, seed):
    rng = np.random.RandomState(seed)
    return dict(teff=rng.uniform(3500, 11500, S), logg=rng.uniform(0.3, 4.7, S),
                feh=rng.uniform(-1.9, -0.1, S), alpha=rng.uniform(0.05, 0.95, S),
                vel=rng.normal(0, 100, S),
                snr=10**rng.uniform(1, np.log10(300), S), seed=seed)


def make_spectra_device(tp, device):
    """Observed spectra synthesised directly in HBM (float64)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(tp['seed']) + 77)
    t = {k: torch.as_tensor(np.asarray(v, dtype=np.float64)).to(device)
         for k, v in tp.items() if k != 'seed'}
    arms = []
    for a in ARMS:
        lam = obs_lam(a)
        wres = 0.5 * sum(arm_def(a)['templ'][:2]) / RESOL / 2.35
        sp0 = synth.spectra_batch(lam, t['teff'], t['logg'], t['feh'],
                                  t['alpha'], vel=t['vel'], wresol=wres,
                                  xp=torch)
        es = sp0 / t['snr'][:, None]
        noise = torch.randn(sp0.shape, dtype=torch.float64, device=device,
                            generator=g)
        spec = sp0 + es * noise
        bad = torch.rand(sp0.shape, device=device, generator=g) < 0.05
        es = torch.where(bad, es * 1e4, es)  # masked: sigma inflated
        arms.append((arm_name(a), lam, spec, es, bad.to(torch.uint8)))
    return arms


def make_spectra_from_library(tp, device, config):
    """Observed spectra drawn from the REGISTERED libraries themselves (float64, in
    HBM): the library's template at the truth parameters, at the truth velocity on
    the arm's pixels -- the `raw_models` of spec_fit.get_chisq(full_output), i.e.
    the library's own evaluator, spline and Doppler factor -- plus noise at the
    truth S/N and 5 % masked pixels.  For evaluators whose templates are not the
    synthetic family's (--evaluator nn: an MLP with random weights): spectra from
    synth.spectra_batch would have nothing to do with such a library and every fit
    would end with a chi^2 warning (round 5: success_frac 0.0 on MLP libraries)."""
    import torch
    from rvspecfit_amd import engine, spec_fit
    g = torch.Generator(device=device)
    g.manual_seed(int(tp['seed']) + 77)
    t = {k: torch.as_tensor(np.asarray(v, dtype=np.float64)).to(device)
         for k, v in tp.items() if k != 'seed'}
    S = t['vel'].shape[0]
    one = [engine.ArmData(arm_name(a), obs_lam(a),
                          torch.ones((S, len(obs_lam(a))), dtype=torch.float64,
                                     device=device),
                          torch.ones((S, len(obs_lam(a))), dtype=torch.float64,
                                     device=device), device=device) for a in ARMS]
    par = torch.stack([t['teff'], t['logg'], t['feh'], t['alpha']], dim=1)
    outp = spec_fit.get_chisq(engine.SpecBatch(one), t['vel'].contiguous(),
                              par.contiguous(), None, None, options=OPTIONS,
                              config=config, full_output=True)
    arms = []
    for a, sp0 in zip(ARMS, outp['raw_models']):
        es = sp0 / t['snr'][:, None]
        noise = torch.randn(sp0.shape, dtype=torch.float64, device=device,
                            generator=g)
        spec = sp0 + es * noise
        bad = torch.rand(sp0.shape, device=device, generator=g) < 0.05
        es = torch.where(bad, es * 1e4, es)
        arms.append((arm_name(a), obs_lam(a), spec, es, bad.to(torch.uint8)))
    return arms


# ------------------------------------------------------------------ CPU leg
def usable_cores():
    """host cores this process can actually run on: the affinity mask, capped by
    the container's CPU quota (cgroup cpu.max / cfs_quota) -- a GPU box may show
    256 CPUs and grant 16 of them; os.cpu_count() workers would then time each
    other's throttling"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(float(q) / float(p))))
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def cpu_worker(args):
    """Runs in a fresh process (never touches the GPU): the oracle -- the CPU
    port of the reference -- on a bounded sample of the same workload."""
    os.environ['OMP_NUM_THREADS'] = '1'
    from oracle import rvs_oracle as orc
    import multiprocessing as mp
    d = dict(np.load(args.cpu_worker))
    n = int(d['n'])
    t0 = time.time()
    dicts = build_library_dicts(args.ccf_every, orc.convolve_vsini_rows)
    tlib = time.time() - t0
    global _W
    _W = dict(libs={k: orc.make_library(v) for k, v in dicts.items()}, d=d)
    ncore = max(1, min(args.cpu_cores or usable_cores(), n))
    one_fn = _cpu_one_process if args.cpu_process else _cpu_one
    # The timed region holds the fits only: the pool is started and every worker
    # has made one untimed call (first-use set-up: library views, the C port's
    # dlopen, scipy imports) before the clock starts.  With one spectrum per
    # worker and the pool start inside the clock, round 2's line understated
    # the CPU six-fold.
    if ncore > 1:
        with mp.get_context('fork').Pool(ncore, initializer=_cpu_warm,
                                         initargs=(bool(args.cpu_process), )
                                         ) as pool:
            pool.map(_cpu_noop, range(4 * ncore), chunksize=1)  # all workers up
            t0 = time.time()
            recs = pool.map(one_fn, range(n), chunksize=1)
            wall = time.time() - t0
    else:
        one_fn(0)
        t0 = time.time()
        recs = [one_fn(i) for i in range(n)]
        wall = time.time() - t0
    t1 = time.time()
    one_fn(0)
    one = time.time() - t1
    print(json.dumps(dict(n=n, wall=wall, cores=ncore, one_spectrum_s=one,
                          lib_s=tlib, recs=[list(map(float, r)) for r in recs])))


def _cpu_warm(process):
    # (an exception in a Pool initializer makes the pool respawn workers for ever:
    # a failing first call is left to the first real task, which reports it)
    try:
        (_cpu_one_process if process else _cpu_one)(0)
    except Exception:   # noqa: BLE001
        pass


def _cpu_noop(i):
    time.sleep(0.01)
    return i


def _cpu_specdata(orc, d, i):
    """spectrum i of the sample as the oracle's SpecData (--workload sdss: on
    its own piece of the lattice)"""
    if 'pieces' in d:
        a0, n = (int(_) for _ in d['pieces'][i])
        return [orc.SpecData(arm_name(a), obs_lam(a)[a0:a0 + n],
                             d['spec_' + a][i][:n], d['espec_' + a][i][:n],
                             badmask=d['bad_' + a][i][:n] != 0) for a in ARMS]
    return [orc.SpecData(arm_name(a), obs_lam(a), d['spec_' + a][i],
                         d['espec_' + a][i], badmask=d['bad_' + a][i] != 0)
            for a in ARMS]


def _cpu_one(i):
    from oracle import rvs_oracle as orc
    libs, d = _W['libs'], _W['d']
    sds = _cpu_specdata(orc, d, i)
    o = orc.ccf_fit(sds, CONFIG, libs)
    vg = np.arange(CONFIG['min_vel'], CONFIG['max_vel'], CONFIG['vel_step0'])
    vs = o['best_vsini']
    rot = None if np.isnan(vs) else (vs, )
    grid = orc.chisq_grid_fast(sds, vg, o['best_par'], rot, OPTIONS, CONFIG,
                               libs)
    s = orc.grid_summary(vg, grid[:, None])
    c = orc.get_chisq_continuum(sds, options=OPTIONS)
    return [o['best_id'], o['best_vel'], s['best_vel'], s['vel_err'],
            s['best_chi']] + list(c['chisq_array'])


def _cpu_one_process(i):
    """oracle vel_fit.process from the (GPU) CCF parameters of spectrum i"""
    from oracle import rvs_oracle as orc
    libs, d = _W['libs'], _W['d']
    sds = _cpu_specdata(orc, d, i)
    names = ['teff', 'logg', 'feh', 'alpha']
    pd0 = {k: float(d['start'][i, j]) for j, k in enumerate(names)}
    pd0['vsini'] = float(d['start'][i, 4])
    cfg = dict(CONFIG)
    cfg.setdefault('max_vsini', 500)
    try:
        r = orc.process(sds, pd0, None, OPTIONS, cfg, libs)
    except RuntimeError:
        # the reference raises on a non finite likelihood (spec_fit.py:963-974: an
        # MLP far outside its training box overflows); the batch path records a
        # status bit for such a spectrum instead
        return [np.nan] * 14
    return [r['vel'], r['vel_err'], r['chisq'], r['vsini']] + \
        [r['param'][k] for k in names] + [r['param_err'][k] for k in names] + \
        [sum(r['nm_nit']), sum(r['nm_nfev'])]


def run_cpu_baseline(arms, n, args, start=None):
    path = '/tmp/rvs_bench_cpu_sample_%d.npz' % os.getpid()
    sample = dict(n=n)
    if start is not None:
        sample['start'] = start[:n]
    for (name, lam, spec, es, bad), a in zip(arms, ARMS):
        sample['spec_' + a] = spec[:n].cpu().numpy()
        sample['espec_' + a] = es[:n].cpu().numpy()
        sample['bad_' + a] = bad[:n].cpu().numpy()
    if SDSS_PIECES is not None:
        sample['pieces'] = SDSS_PIECES[:n]
    np.savez(path, **sample)
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-worker', path,
           '--ccf-every', str(args.ccf_every), '--cpu-cores',
           str(args.cpu_cores), '--workload', args.workload, '--evaluator',
           args.evaluator]
    if getattr(args, 'grid', ''):
        cmd += ['--grid', args.grid]
    cmd += ['--npoly', str(OPTIONS['npoly'])]
    if start is not None:
        cmd.append('--cpu-process')
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=1500)
    os.unlink(path)
    if out.returncode != 0:
        raise RuntimeError('cpu baseline failed: ' + out.stderr[-2000:])
    return json.loads(out.stdout.strip().splitlines()[-1])


# ------------------------------------------------------------------ launcher
RANK_ENV = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT',
            'HSA_ENABLE_IPC_MODE_LEGACY')


def launch_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: start one FRESH
    child process per rank (this very command line, with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set as torch.distributed.run would set them), relay
    rank 0's result line and return non-zero if any rank failed.  The parent
    imports neither torch nor the HIP library and never re-executes itself: a
    process that has initialised the GPU must not be replaced by another.  One
    process per GPU is the reference's own model of parallelism (one worker
    per spectrum, desi/desi_fit.py:1215-1218, 1475-1479)."""
    import socket
    if 'MASTER_PORT' in os.environ:
        port = int(os.environ['MASTER_PORT'])
    else:
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get(
                       'HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen(
            [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
            stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    # rank 0's stdout is this command's stdout; the other ranks' goes to stderr
    import threading

    def relay():
        for ln in procs[0].stdout:
            # stdout carries the result line only (gloo prints its connection
            # notes on stdout)
            f = sys.stdout if ln.lstrip().startswith('{') else sys.stderr
            f.write(ln)
            f.flush()
    th = threading.Thread(target=relay, daemon=True)
    th.start()
    rc = 0
    deadline = time.time() + 3600
    live = list(procs)
    while live:
        for p in list(live):
            c = p.poll()
            if c is None:
                continue
            live.remove(p)
            if c != 0:
                rc = rc or (c if c > 0 else 1)
                # one rank down: the others would wait in a collective for ever
                for q in live:
                    q.terminate()
        if time.time() > deadline:
            for q in live:
                q.kill()
            return rc or 1
        time.sleep(0.05)
    th.join(timeout=10)
    return rc


def dry_launch():
    """What a rank was started with, and the process group those variables
    form (gloo, CPU only): the launch path without the GPU."""
    info = {k: os.environ.get(k) for k in RANK_ENV}
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if os.environ.get('RVS_BENCH_FAIL_RANK') == os.environ.get('RANK', '0'):
        sys.exit(3)   # (tests: a failing rank must fail the whole command)
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group('gloo')
        got = [None] * world
        dist.all_gather_object(got, (dist.get_rank(), os.getpid()))
        info['n_gpus'] = dist.get_world_size()
        info['ranks'] = [g[0] for g in got]
        info['pids'] = [g[1] for g in got]
        dist.barrier()
        dist.destroy_process_group()
    else:
        info['n_gpus'] = 1
    if int(os.environ.get('RANK', '0')) == 0:
        print(json.dumps(info), flush=True)


# ------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--spectra', type=int, default=10000,
                    help='spectra per GPU per step')
    ap.add_argument('--ccf-every', type=int, default=64,
                    help='grid subsampling of the CCF set: 64 -> T=76, 9 -> T=534')
    ap.add_argument('--cpu-sample', type=int, default=0,
                    help='spectra of the CPU baseline sample (0: 8 per host '
                         'core, at most 512)')
    ap.add_argument('--cpu-cores', type=int, default=0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--refine', action='store_true',
                    help='also run the _minimum_sampler refinement (add-on)')
    ap.add_argument('--cpu-worker', type=str, default=None)
    ap.add_argument('--resolution-matrix', action='store_true',
                    help='add-on workload: every spectrum carries an 11-diagonal '
                         'resolution matrix (DESI --resolution_matrix mode, A9)')
    ap.add_argument('--process', type=int, default=0,
                    help='add-on: also run vel_fit.process (Nelder-Mead + '
                         'Hessian, SURVEY 8(f) rank 1) on this many spectra '
                         'starting from the CCF parameters; reported under '
                         '"process", never part of `value`')
    ap.add_argument('--process-cpu-sample', type=int, default=8)
    ap.add_argument('--process-no-bfgs', action='store_true',
                    help='--process / --desi-file run the reference\'s default '
                         '(utils.py:26 second_minimizer = True: BFGS after '
                         'Nelder-Mead); this switches the second minimiser off '
                         '(the add-on figure)')
    ap.add_argument('--process-bfgs', action='store_true',
                    help='(the default since round 6; kept so that older command '
                         'lines still parse)')
    ap.add_argument('--desi-file', type=int, default=0,
                    help='add-on: write the first N spectra as a DESI coadd FITS '
                         'file and run the driver (desi_fit.proc_desi) on it: '
                         'read -> select -> condition -> CCF + process + '
                         'continuum -> RVTAB/RVMOD')
    ap.add_argument('--desi-nfiles', type=int, default=1,
                    help='with --desi-file: this many copies of the file are '
                         'processed as one group (desi_fit.proc_desi_group, what '
                         'proc_many does with files_per_batch)')
    ap.add_argument('--desi-files-per-batch', type=int, default=8,
                    help='with --desi-nfiles: files fitted together in one GPU '
                         'batch (proc_many files_per_batch)')
    ap.add_argument('--desi-workers', type=int, default=1,
                    help='with --desi-nfiles: worker processes sharing the GPU '
                         '(proc_many nthreads); the synthetic template libraries '
                         'are written to disk for them')
    ap.add_argument('--cpu-process', action='store_true',
                    help='(cpu worker) run the oracle process stage')
    ap.add_argument('--npoly', type=int, default=10,
                    help='continuum basis size (the reference\'s tests and the WEAVE '
                         'driver run 15); the flop model follows it')
    ap.add_argument('--workload', choices=['desi', 'cfg2', 'sdss'], default='desi',
                    help='desi: BASELINE configs[2] (3 arms); cfg2: configs[1] '
                         '(1 arm, 2001 px, N_fft 4096)')
    ap.add_argument('--evaluator', choices=['polylinear', 'nn', 'tri'],
                    default='polylinear',
                    help='nn: BASELINE configs[3], MLP template evaluator on MFMA; '
                         'tri: the same grid nodes as a Delaunay library '
                         '(spec_inter.TriInterp: 31 104 simplices for the 7^4 grid), '
                         'find_simplex through the bucket grid')
    ap.add_argument('--grid', type=str, default='',
                    help='template grid nodes per dimension "nteff,nlogg,nfeh,'
                         'nalpha" (default 7,7,7,7 = 60 MB/arm, Infinity-Cache '
                         'resident); 40,11,8,5 = 17 600 templates, 440 MB/arm: a '
                         'library of realistic size, dimensions of different '
                         'length, gathers served from HBM')
    ap.add_argument('--per-spectrum-templates', action='store_true',
                    help='build one template per spectrum (rounds 1-2) instead of '
                         'one per CCF node shared by the spectra that selected it; '
                         'same records bit for bit')
    ap.add_argument('--dump-records', type=str, default='',
                    help='rank 0 saves the gathered [n_gpus * spectra, 16] result '
                         'table of the last step as .npy (tests)')
    ap.add_argument('--seed-rank', type=int, default=-1,
                    help='generate the spectra rank R of a multi-rank run would '
                         'generate (tests: a 1-rank run that reproduces one shard)')
    ap.add_argument('--dry-launch', action='store_true',
                    help='with --gpus N: every rank prints the environment it '
                         'was started with and exits (no GPU, no torch)')
    args = ap.parse_args()
    global ARMS, EVALUATOR, GRID_KW
    if args.workload == 'cfg2':
        ARMS = ('c', )
    if args.workload == 'sdss':
        ARMS = ('s', )
    EVALUATOR = args.evaluator
    assert 1 <= args.npoly <= 16, '--npoly: 1..16'
    OPTIONS['npoly'] = args.npoly
    if args.grid:
        g = [int(_) for _ in args.grid.split(',')]
        assert len(g) == 4 and min(g) >= 2, '--grid needs four sizes >= 2'
        GRID_KW = dict(nteff=g[0], nlogg=g[1], nfeh=g[2], nalpha=g[3])
    if args.cpu_worker:
        return cpu_worker(args)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process starts
        # the N ranks itself and never touches the GPU
        sys.exit(launch_ranks(args.gpus))
    if int(os.environ.get('WORLD_SIZE', '1')) != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%s: refusing to '
                         'report a line for a run that is not the one asked '
                         'for\n' % (args.gpus, os.environ.get('WORLD_SIZE')))
        sys.exit(2)
    if args.dry_launch:
        return dry_launch()

    import torch
    import torch.distributed as dist
    from rvspecfit_amd import _lib, engine, pipeline, spec_inter
    from rvspecfit_amd import dist as rdist
    from rvspecfit_amd.library import TemplateLibrary

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    _lib.require_gpu()
    if os.environ.get('RVS_SHARE_GPU'):
        local = 0   # functional multi-rank run on ONE GPU (with RVS_DIST_BACKEND=gloo)
    elif local >= torch.cuda.device_count():
        sys.stderr.write('bench.py: rank %d has no GPU of its own (%d visible); '
                         'set RVS_SHARE_GPU=1 RVS_DIST_BACKEND=gloo for a '
                         'functional run on one GPU\n'
                         % (rank, torch.cuda.device_count()))
        sys.exit(2)
    torch.cuda.set_device(local)
    try:
        rdist.init_from_env(backend='nccl')
    except rdist.DistInitError as e:
        # no retry, no other backend: this process has initialised the GPU.  The
        # parent (launch_ranks / torch.distributed.run) relays the exit code.
        sys.stderr.write('bench.py: %s\n' % e)
        sys.exit(4)
    if world > 1:
        # what the line reports is what torch.distributed actually formed
        world = dist.get_world_size()
        rank = dist.get_rank()
    dev = torch.device('cuda', local)
    _lib.lib()
    S = args.spectra

    def gpu_convolve(lam, templ, vsini):
        t = torch.as_tensor(np.ascontiguousarray(templ)).to(dev)
        v = torch.as_tensor(np.ascontiguousarray(vsini)).to(dev)
        return engine.convolve_vsini(lam, t, v).cpu().numpy()

    t_setup = time.time()
    dicts = build_library_dicts(args.ccf_every, gpu_convolve,
                                device=dev if args.grid else None)
    for name, d in dicts.items():
        spec_inter.register_library(TemplateLibrary(name, d, device=dev),
                                    CONFIG['template_lib'])
    Tccf = dicts[arm_name(ARMS[0])]['ccf_fft'].shape[0]
    nfft = int(dicts[arm_name(ARMS[0])]['ccf_npoints'])
    tp = truth_params(S, seed=3 + 1000 * (args.seed_rank if args.seed_rank >= 0
                                          else rank))
    arms = make_spectra_device(tp, dev) if EVALUATOR != 'nn' else \
        make_spectra_from_library(tp, dev, CONFIG)
    if args.workload == 'sdss':
        # every spectrum keeps its own piece of the lattice: its pixels move to the
        # front of its row, the rest is padding (engine.ArmData, grid sets)
        global SDSS_PIECES
        SDSS_PIECES = sdss_pieces(S, int(tp['seed']))
        name, lam, sp, es, bad = arms[0]
        a0 = torch.as_tensor(SDSS_PIECES[:, 0]).to(dev)
        npx = int(SDSS_PIECES[:, 1].max())
        col = (a0[:, None] + torch.arange(npx, device=dev)[None, :]).clamp_(
            max=sp.shape[1] - 1)
        arms = [(name, lam, sp.gather(1, col), es.gather(1, col),
                 bad.gather(1, col))]
        grids = [lam[a:a + n] for a, n in SDSS_PIECES]
        batch = engine.SpecBatch([engine.ArmData(
            name, grids, arms[0][2], arms[0][3], arms[0][4], device=dev,
            grid_id=np.arange(S, dtype=np.int32))])
    else:
        batch = engine.SpecBatch([engine.ArmData(n, lam, sp, es, bad, device=dev)
                                  for n, lam, sp, es, bad in arms])
    if args.resolution_matrix:
        # per-spectrum Gaussian rows, sigma 0.45-0.65 px-units of 0.8 A, 11 taps,
        # rows normalised (what desi_fit.construct_resolution_sparse_matrix
        # hands over); built directly on the device
        g = torch.Generator(device=dev)
        g.manual_seed(991 + rank)
        for a in batch.arms:
            sig = 0.45 + 0.2 * torch.rand((S, 1, 1), device=dev, generator=g,
                                          dtype=torch.float64)
            d = torch.arange(-5, 6, device=dev, dtype=torch.float64)[None, None]
            k = torch.arange(a.npix, device=dev)[None, :, None]
            t = torch.exp(-0.5 * (d / (sig / 0.8 * 1.0))**2).expand(
                S, a.npix, 11).clone()
            q = k + d.long()
            t = torch.where((q >= 0) & (q < a.npix), t, torch.zeros_like(t))
            t = t / t.sum(dim=2, keepdim=True)
            a.resol = dict(taps=t.contiguous(), nd=11, stride=a.npix * 11,
                           unit=t.sum(dim=2).contiguous())
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    def step():
        for a in batch.arms:
            a._work.clear()  # per-spectrum preparation belongs to the step
        rec = pipeline.fit_batch(batch, CONFIG, options=OPTIONS,
                                 refine=args.refine,
                                 share_templates=not args.per_spectrum_templates)
        # the only collective of the path: gather of the result records
        return rdist.gather_records(rec, world * S, rank, world)

    # the warm-up steps run exactly what the timed steps run, per-kernel event
    # timers included: on a freshly started machine the first use of a code
    # path pages it in from disk, and with the timers switched on only for the
    # timed steps that cost the first of them 15 % (40.3k vs 43.9k spectra/s)
    engine.KTIMERS = {}
    for _ in range(args.warmup):
        rec = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    engine.KTIMERS = {}
    stage = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rec = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kt = engine.ktimers_summary()
    engine.KTIMERS = None
    if world > 1:
        tdev = dev if dist.get_backend() == 'nccl' else 'cpu'
        tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # one extra, untimed, instrumented step for the per-stage breakdown
    timers = {}
    for a in batch.arms:
        a._work.clear()
    pipeline.fit_batch(batch, CONFIG, options=OPTIONS, timers=timers,
                       share_templates=not args.per_spectrum_templates)
    torch.cuda.synchronize()
    stage = {k: v[0].elapsed_time(v[1]) for k, v in timers.items()}

    # ---- the polylinear gather on its own (A3), at every spectrum's OWN
    # parameters: the step above evaluates templates at the CCF nodes (38
    # distinct cells, cache resident whatever the library size); the optimiser
    # stage and a first-guess grid evaluate them anywhere in the grid, i.e.
    # 16 rows per job scattered over the whole library
    gather = None
    if EVALUATOR in ('polylinear', 'tri') and rank == 0:
        # (rows blended per template: the 2^4 float32 vertex rows of a grid cell, or
        # the 5 float64 rows of a Delaunay simplex behind find_simplex)
        rows_b = 16 * 4 if EVALUATOR == 'polylinear' else 5 * 8
        ptrue = torch.as_tensor(np.stack(
            [tp[k] for k in ('teff', 'logg', 'feh', 'alpha')], axis=1)).to(dev)
        gb, gms = 0.0, 0.0
        for a in ARMS:
            lib = spec_inter.get_libs([arm_name(a)], CONFIG)[arm_name(a)]
            lib.eval_batch(ptrue)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                lib.eval_batch(ptrue)
            e1.record()
            torch.cuda.synchronize()
            gms += e0.elapsed_time(e1) / 3
            # algorithmic bytes: 16 float32 rows in, one float64 template out
            gb += S * lib.ntp * (rows_b + 8)
        gather = dict(ms_per_batch=round(gms, 3),
                      alg_GBps=round(gb / (gms * 1e-3) / 1e9, 1),
                      frac_of_hbm_peak=round(gb / (gms * 1e-3) / 1e9
                                             / HBM_PEAK_GBS, 4),
                      library_MB_per_arm=[round(
                          dicts[arm_name(a)]['dats'].shape[0]
                          * len(dicts[arm_name(a)]['lam'])
                          * (4 if EVALUATOR == 'polylinear' else 8) / 1e6, 1)
                          for a in ARMS],
                      note=('rvs_template_polylinear at the %d spectra\'s own '
                            '(random in-grid) parameters, all arms; bytes = 16 '
                            'float32 rows read + one float64 template written '
                            'per job' % S) if EVALUATOR == 'polylinear' else
                      ('rvs_template_tri_buckets (find_simplex through the bucket '
                       'grid + blend) at the %d spectra\'s own parameters, all arms; '
                       'bytes = 5 float64 rows read + one template written per job; '
                       '%d simplices' % (S, dicts[arm_name(ARMS[0])][
                           'simplices'].shape[0])))

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    if args.dump_records:
        np.save(args.dump_records, rec.cpu().numpy())

    value = world * S * args.steps / dt
    # ---- roofline of the dominant kernel -------------------------------
    # algorithmic bytes: the CCF template block streamed once
    # per spectrum: sum_arm T*(nfft/2+1)*16 B*2 (+ the spectrum's own S*,V*)
    n2 = nfft // 2 + 1
    b_ccf_unit = len(ARMS) * Tccf * n2 * 32
    nl, ms, units = kt.get('ccf_xcorr', (0, 0.0, 0))
    # `units` counts spectrum-arm launches units (n spectra per arm launch)
    ccf_bytes = units / len(ARMS) * b_ccf_unit
    ccf_gbs = ccf_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    nl2, ms2, units2 = kt.get('chisq_grid', (0, 0.0, 0))
    # (a grid set: the spectra's own pixel counts, not the padded row length)
    npix_tot = sum(float(a.npix_g[a.grid_id_host].mean()) if a.G > 1 else a.npix
                   for a in batch.arms)
    ntp_tot = sum(len(dicts[arm_name(a)]['lam']) for a in ARMS)
    # chi^2 grid: spectrum terms (16 B/px) + spline records (32 B/knot) + out
    b_grid_unit = npix_tot * 16 + ntp_tot * 32 + 400 * 8
    # (the timer counts job-velocities per arm launch; a contract unit is 400
    # velocities on every arm, refinement rounds add shorter grids)
    units2 = units2 / 400.0
    grid_gbs = (units2 / len(ARMS)) * b_grid_unit / (ms2 * 1e-3) / 1e9 if ms2 else 0
    P_ = OPTIONS['npoly']
    nsum_ = P_ * (P_ + 1) // 2 + P_
    # (--resolution-matrix: + the 11-tap band on the resampled template, one FMA
    # per tap and pixel-velocity)
    flop_grid_unit = 400 * npix_tot * (2 * nsum_ + 40 +
                                       (22 if args.resolution_matrix else 0))
    grid_tflops = (units2 / len(ARMS)) * flop_grid_unit / (ms2 * 1e-3) / 1e12 if ms2 else 0
    # ---- roofline of the DOMINANT kernel: the fp64 chi^2 grid ----------------
    # algorithmic flops per spectrum: Nv * sum_arm npix * (2*65 + 40) at npoly 10
    # (65 FMAs of the normal-equation sums per pixel-velocity + 40 for the spline
    # value, weights, products); one launch = one rvs_chisq_grid
    # call over one arm (full-wave kernel + packed-wave kernel)
    FP64_PEAK_TF = 78.6   # datasheet fp64 vector rate (256 CUs x 128 flop/clk x 2.4 GHz)

    grid_name = 'x'.join(str(GRID_KW[k]) for k in ('nteff', 'nlogg', 'nfeh',
                                                    'nalpha'))
    tkey = traffic_key(args, S, Tccf, nfft, grid_name)

    def pmc_traffic(kernel):
        """HBM bytes per launch of `kernel` for THIS configuration, from the
        committed rocprofv3 counter passes of this build (FETCH_SIZE x 2 +
        WRITE_SIZE, separate runs, tools/perf/pmc_traffic.sh): the entry of
        profiles/rNN_pmc_traffic.json (the newest round's) whose key is this run's configuration, or
        None when that configuration has not been profiled -- a figure measured
        on another workload is never stamped on this line"""
        f = profile_file('pmc_traffic.json')
        try:
            d = json.load(open(f))[tkey][kernel]
            return d['hbm_bytes_per_launch'], \
                'profiles/%s[%s] (%s)' % (os.path.basename(f), tkey, d['source'])
        except Exception:
            return None, None

    def sq_counters(kernel):
        """SQ / GRBM counters per launch (tools/perf/xc_counters.sh), default
        configuration only; None otherwise"""
        f = profile_file('sq_counters.json')
        try:
            d = json.load(open(f))
            if d.get('traffic_key') != tkey:
                return None
            for k, v in d['kernels'].items():   # ('void name<..>' for templates)
                if kernel in k:
                    return v
            return None
        except Exception:
            return None
    tr_grid, src_grid = pmc_traffic('chisq_grid')
    tr_ccf, src_ccf = pmc_traffic('ccf_xcorr')
    roof = dict(bound='fp64_valu', kernel='chisq_grid_kernel',
                timed='rvs_chisq_grid call = chisq_grid_kernel<10,false> (full '
                      'waves) + <10,true> (packed left-over velocities, side '
                      'stream, joined before the call returns), HIP events '
                      'on the launch stream around the call',
                achieved=round(grid_tflops, 2), peak=FP64_PEAK_TF,
                unit='TFLOP/s', frac=round(grid_tflops / FP64_PEAK_TF, 4),
                traffic=tr_grid, traffic_source=src_grid,
                flop_per_spectrum=flop_grid_unit,
                bytes_per_spectrum=b_grid_unit,
                avg_launch_ms=round(ms2 / max(nl2, 1), 3), launches=nl2,
                share_of_step=round(ms2 / args.steps / (dt / args.steps * 1e3), 3),
                npoly=OPTIONS['npoly'],
                note='fp64 vector-ALU bound (0.07 TB/s algorithmic): flops = '
                     '2 (P (P + 1) / 2 + P) + 40 per pixel-velocity (170 at npoly 10; '
                     '+ 22 for the 11-tap band with --resolution-matrix); peak = datasheet fp64 vector rate (a '
                     'pure v_fma_f64 loop sustained 70.6 TF on this chip in '
                     'round 2, tools/perf/ubench.hip -- not measured by this run)')
    roof_ccf = dict(bound='hbm', kernel='ccf_xcorr_ws_kernel (nfft 8192: one persistent '
                    'block per spectrum, wave-specialised) / ccf_xcorr_kernel',
                    timed='rvs_ccf_xcorr call = ccf_rfft_kernel + the cross-correlation '
                          'kernel (HIP events on the launch stream)',
                    achieved=round(ccf_gbs, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=round(ccf_gbs / HBM_PEAK_GBS, 4),
                    traffic=tr_ccf, traffic_source=src_ccf,
                    bytes_per_spectrum=b_ccf_unit,
                    avg_launch_ms=round(ms / max(nl, 1), 3), launches=nl,
                    share_of_step=round(ms / args.steps / (dt / args.steps * 1e3), 3),
                    note='BYTE MODEL, not HBM utilisation: algorithmic bytes = CCF '
                         'template block streamed once per spectrum (SURVEY 8(d) '
                         'D3); the block is shared by every spectrum, so it is '
                         'served by L2 / Infinity Cache and `achieved` can exceed '
                         'what HBM can deliver -- see counter_backed')
    # what the counters say bounds the kernel (committed passes of this build on
    # this configuration; null when not profiled)
    sq = sq_counters('ccf_xcorr_ws_kernel') or sq_counters('ccf_xcorr_kernel')
    avg_ms = ms / max(nl, 1)
    cb_ = dict(real_hbm_GBps=None, real_hbm_frac_of_peak=None, l2_to_l1_TBps=None,
               valu_busy=None, lds_busy=None, source=None)
    if tr_ccf and avg_ms > 0:
        # one rvs_ccf_xcorr call = one launch of ccf_xcorr_kernel (+ rfft)
        cb_['real_hbm_GBps'] = round(tr_ccf / (avg_ms * 1e-3) / 1e9, 1)
        cb_['real_hbm_frac_of_peak'] = round(cb_['real_hbm_GBps'] / HBM_PEAK_GBS, 4)
        cb_['source'] = src_ccf
    if sq:
        # vector-memory read instructions x 64 lanes x 16 B / kernel time
        cb_['l2_to_l1_TBps'] = sq.get('l2_to_l1_TBps')
        cb_['valu_busy'] = sq.get('valu_busy')
        cb_['lds_busy'] = sq.get('lds_busy')
        cb_['source'] = (cb_['source'] or '') + ' + profiles/' + os.path.basename(
            profile_file('sq_counters.json'))
    roof_ccf['counter_backed'] = cb_
    kernels = {
        'ccf_xcorr': dict(ms_per_step=round(ms / args.steps, 2),
                          alg_GBps=round(ccf_gbs, 1)),
        'chisq_grid': dict(ms_per_step=round(ms2 / args.steps, 2),
                           alg_GBps=round(grid_gbs, 2),
                           fp64_TFLOPs=round(grid_tflops, 2)),
    }
    if gather is not None:
        kernels['template_polylinear' if EVALUATOR == 'polylinear'
                else 'template_tri'] = gather
    if 'ccf_preprocess' in kt:
        kernels['ccf_preprocess'] = dict(
            ms_per_step=round(kt['ccf_preprocess'][1] / args.steps, 2))
    if 'template_nn' in kt:
        # f32 MLP on v_mfma_f32_32x32x2_f32: 2*sum(K*N) flop per spectrum-arm
        nl3, ms3, units3 = kt['template_nn']
        fl = 0.0
        for a in ARMS:
            dims = dicts[arm_name(a)]['nn_dims']
            fl += 2.0 * sum(int(dims[i]) * int(dims[i + 1]) for i in range(5))
        tf = (units3 / len(ARMS)) * fl / (ms3 * 1e-3) / 1e12 if ms3 else 0
        kernels['template_nn'] = dict(
            ms_per_step=round(ms3 / args.steps, 2), f32_mfma_TFLOPs=round(tf, 2),
            mfma_peak_TFLOPs=157.3, mfma_frac=round(tf / 157.3, 4))

    # ---- CPU baseline + parity on the sample ---------------------------
    cpu = None
    parity = None
    # the CPU leg is timed on rank 0 at N = 1 only
    if not args.no_cpu_baseline and not args.resolution_matrix and world == 1:
        ncpu = args.cpu_cores or usable_cores()
        n = min(args.cpu_sample or min(512, 8 * ncpu), S)
        cb = run_cpu_baseline(arms, n, args)
        per_core = cb['n'] / cb['wall'] / cb['cores']
        cpu = dict(value=round(cb['n'] / cb['wall'], 3), unit='spectra/s',
                   cores=cb['cores'], kind='port',
                   sample='%d of the %d spectra of rank 0 (%.1f per core), oracle '
                          '(numpy/scipy + C port of the reference) CCF + '
                          '400-velocity chi^2 grid + continuum chi^2, '
                          'process-parallel over %d host cores, OMP_NUM_THREADS=1; '
                          'pool start and one warm-up call per worker are outside '
                          'the clock' % (cb['n'], S, cb['n'] / cb['cores'],
                                         cb['cores']),
                   spectra_per_s_per_core=round(per_core, 4),
                   one_spectrum_seconds_1core=round(cb['one_spectrum_s'], 4),
                   # ~1 when the workers scale; < 1: the cores share memory
                   # bandwidth / boost clocks when all of them run
                   per_core_rate_x_one_spectrum_seconds=round(
                       per_core * cb['one_spectrum_s'], 3))
        g = rec[:n].cpu().numpy()
        o = np.array(cb['recs'])
        same = (g[:, 0] == o[:, 0])
        parity = dict(n=n, best_id_equal=int(same.sum()),
                      max_abs_dvrad_ccf=float(np.abs(g[:, 1] - o[:, 1]).max()),
                      max_abs_drv_same_template=float(
                          np.abs(g[same, 7] - o[same, 2]).max()) if same.any() else None,
                      # -2 log L = log det + 2 sum log e + residual can pass
                      # through zero, so the difference is scaled by
                      # max(|chi|, total number of pixels) (the residual term is
                      # of the order of the pixel count)
                      max_rel_dchi_same_template=float(
                          (np.abs(g[same, 11] - o[same, 4]) /
                           np.maximum(np.abs(o[same, 4]), npix_tot)).max())
                      if same.any() else None,
                      max_abs_dchi_same_template=float(
                          np.abs(g[same, 11] - o[same, 4]).max())
                      if same.any() else None)

    proc = None
    if args.process > 0:
        proc = run_process_addon(batch, rec, arms, args, dev)
    desi = None
    if args.desi_file > 0 and rank == 0 \
            and args.workload == 'desi':
        desi = run_desi_addon(arms, args, dev, dicts)

    # SURVEY 8(d) D3 end-to-end byte model: B_alg = spectrum terms + polylinear
    # gather + CCF template block + outputs, per spectrum
    b_alg = npix_tot * 16 + (16 * 4 * ntp_tot if EVALUATOR == 'polylinear' else
                             (5 * 8 * ntp_tot if EVALUATOR == 'tri' else 0)) \
        + b_ccf_unit + 4096
    line = dict(
        metric='spectra/sec (CCF+chi2 grid) DESI 3-arm' if args.workload == 'desi'
        else ('spectra/sec (CCF+chi2 grid) 1 arm, every spectrum on its own '
              'wavelength grid (SDSS-style)' if args.workload == 'sdss' else
              'spectra/sec (CCF+chi2 grid) 1 arm 4000-5000 A (BASELINE configs[1])'),
        value=round(value, 1), unit='spectra/s', n_gpus=world, steps=args.steps,
        warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 2),
        higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f64',
        data='synthetic',
        config=dict(workload='%s, %d spectra per GPU per step, %s evaluator, %s '
                             'grid, T=%d CCF templates, N_fft=%d, 400-velocity '
                             'chi2 grid, npoly=%d' % (
                                 'DESI b/r/z 3-arm (2751/2326/2881 px) (BASELINE '
                                 'configs[%d])' % (3 if EVALUATOR == 'nn' else 2)
                                 if args.workload == 'desi' else
                                 ('1 arm on the SDSS log-lambda lattice, %d grids '
                                  'of %d-%d px (one per spectrum; add-on workload)'
                                  % (S, int(SDSS_PIECES[:, 1].min()),
                                     int(SDSS_PIECES[:, 1].max()))
                                  if args.workload == 'sdss' else
                                  '1 arm 4000-5000 A 2001 px (BASELINE configs[1])'),
                                 S, EVALUATOR,
                                 'x'.join(str(GRID_KW[k]) for k in (
                                     'nteff', 'nlogg', 'nfeh', 'nalpha')),
                                 Tccf, nfft, OPTIONS['npoly']),
                    spectra_per_gpu=S, npoly=OPTIONS['npoly'], ccf_templates=Tccf, nfft=nfft,
                    refine=bool(args.refine),
                    resolution_matrix=bool(args.resolution_matrix),
                    templates='one per spectrum' if (
                        args.per_spectrum_templates or not node_templates(Tccf, S)) else
                    'one per CCF node (%d per arm), built every step, shared by '
                    'the spectra that selected the node -- what the reference\'s '
                    'getCurTempl / spline caches do (spec_fit.py:357-407, '
                    '902-910); records identical to per-spectrum templates' % Tccf,
                    traffic_key=tkey,
                    parallelism='spectra-sharded x%d' % world),
        roofline=roof, roofline_ccf=roof_ccf,
        b_alg=dict(bytes_per_spectrum=b_alg,
                   achieved_GBps=round(value / world * b_alg / 1e9, 1),
                   frac_of_hbm_peak=round(value / world * b_alg / 1e9
                                          / HBM_PEAK_GBS, 4),
                   note='SURVEY 8(d) D3 byte model of the WHOLE path per GPU (it '
                        'counts a 16-row gather per spectrum; with one template '
                        'per CCF node the gather is done once per node); the '
                        'path is fp64-compute-side under it (see roofline)'),
        cpu_baseline=cpu,
        stage_ms=stage_round(stage),
        kernels=kernels, parity_sample=parity, setup_s=round(t_setup, 1))
    # what a reader of an N-GPU line needs to hold it against the N = 1 line: the
    # value per GPU (weak scaling: every rank runs the N = 1 workload) and the
    # process group torch.distributed actually formed (backend 'nccl' = RCCL)
    line['per_gpu_value'] = round(value / world, 1)
    line['dist'] = dict(
        world_size=world, backend=dist.get_backend() if world > 1 else None,
        launched_as=int(os.environ.get('WORLD_SIZE', '1')),
        collective='all_gather_into_tensor of the [S, %d] float64 result records '
                   '(rvspecfit_amd/dist.py), inside the timed step'
                   % len(pipeline.RECORD_FIELDS) if world > 1 else None)
    if proc is not None:
        line['process'] = proc
    if desi is not None:
        line['desi_file'] = desi
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def objective_model(ntp, npix, npoly, nvert, ktaps, kind='regulargrid'):
    """Algorithmic fp64 flops and gathered bytes of ONE (job, arm) evaluation of
    the optimiser's objective (csrc/objective.hip), per phase:
      gather   ntp (2 nvert + 35)        nvert-row blend (one FMA per row) + exp
      fir      ntp 2 ktaps               rotational broadening, ktaps = 2 kmax + 1
      spline   ntp 20                    right-hand sides, two sweeps, corrections
      model    npix 25                   spline value at the pixel, units of sigma
      normal   npix (2 (P(P+1)/2 + P) + P + 3)   the P(P+3)/2 sums
      resid    npix (2 P + 3)            explicit residual norm
      solve    P^3 / 3 + 2 P^2           Cholesky + two triangular solves
    bytes: the nvert float32 vertex rows (what comes out of the Infinity Cache /
    HBM per evaluation; the per-arm constants -- spline factors, basis, pixel
    grid -- and the spectrum's 3 npix doubles are L2 traffic and not counted)."""
    P = npoly
    fl = dict(gather=ntp * (2 * nvert + 35) if kind == 'regulargrid' else 0,
              fir=ntp * 2 * ktaps, spline=ntp * 20, model=npix * 25,
              normal=npix * (2 * (P * (P + 1) // 2 + P) + P + 3),
              resid=npix * (2 * P + 3), solve=P ** 3 // 3 + 2 * P * P)
    by = nvert * ntp * 4 if kind == 'regulargrid' else ntp * 8
    return fl, by


def profile_file(suffix):
    """the newest round's committed counter file profiles/rNN_<suffix>"""
    import glob
    c = sorted(glob.glob(os.path.join(REPO, 'profiles', 'r[0-9][0-9]_' + suffix)))
    return c[-1] if c else os.path.join(REPO, 'profiles', 'r00_' + suffix)


def objective_counters(grid_name):
    """SQ / TCC counters of objective_kernel<10> from the committed counter passes
    of this build (tools/perf/obj_counters.sh on tools/perf/obj_bench: 9000 jobs x
    3 DESI arms at random in-grid parameters, the library of this grid) -- not
    measured by this run; None for a grid that has not been profiled."""
    f = profile_file('obj_counters.json')
    try:
        return json.load(open(f))[grid_name]
    except Exception:
        return None


def process_roofline(sub, r, out, tm, dt, dt1):
    """Efficiency of the optimiser stage: objective evaluations per second and what
    they amount to against the fp64 vector peak and the memory system (the stage is
    bound by objective_kernel: one block per CU)."""
    from rvspecfit_amd import spec_inter
    FP64_PEAK_TF = 78.6
    evals = int(r['objective_evals'])
    vs = r['vsini'] if 'vsini' in r else None
    fl_tot, by_tot, per_arm = 0, 0, {}
    for a in sub.arms:
        lib = spec_inter.interp_cache.registered[(CONFIG['template_lib'], a.name)]
        kind = 'regulargrid' if (EVALUATOR == 'polylinear' and
                                 lib.kind == 'regulargrid') else 'from_template'
        nvert = 2 ** lib.ndim if kind == 'regulargrid' else 1
        kt = 1
        if vs is not None:   # taps of the fitted vsini, mean over the spectra
            import torch
            R = (torch.nan_to_num(vs.double(), nan=0.0) / 299792.458) / lib.lnstep
            kt = float((2 * torch.ceil(R + 1) + 1).mean())
        npx = float(a.npix_g[a.grid_id_host].mean()) if a.G > 1 else a.npix
        fl, by = objective_model(lib.ntp, npx, OPTIONS['npoly'], nvert, kt, kind)
        per_arm[a.name] = dict(ntp=lib.ntp, npix=round(npx, 1), nvert=nvert,
                               taps=round(kt, 1), flops=int(sum(fl.values())),
                               flops_by_phase={k: int(v) for k, v in fl.items()},
                               gathered_bytes=by)
        fl_tot += sum(fl.values())
        by_tot += by
    nm_s = tm.get('neldermead', 0.0)
    tf = evals * fl_tot / dt / 1e12
    grid_name = 'x'.join(str(GRID_KW[k]) for k in ('nteff', 'nlogg', 'nfeh', 'nalpha'))
    return dict(
        kernel='objective_kernel<%d,%s>' % (OPTIONS['npoly'],
                                            'false' if EVALUATOR == 'polylinear' else 'true'),
        bound='latency / fp64 VALU issue at two waves per SIMD (one 155-KB-LDS block '
              'per CU); not HBM',
        unit='TFLOP/s', achieved=round(tf, 2), peak=FP64_PEAK_TF,
        frac=round(tf / FP64_PEAK_TF, 4),
        evaluations=evals, arm_evaluations=evals * len(sub.arms),
        evaluations_per_s=round(evals / dt, 0),
        flops_per_evaluation=int(fl_tot), gathered_bytes_per_evaluation=int(by_tot),
        gather_GBps=round(evals * by_tot / dt / 1e9, 1),
        gather_frac_of_hbm_peak=round(evals * by_tot / dt / 1e9 / HBM_PEAK_GBS, 4),
        us_per_arm_evaluation_per_cu=round(
            (nm_s if nm_s else dt) * 256 / max(1, evals * len(sub.arms)) * 1e6 *
            (dt / dt1 if nm_s else 1.0), 2),
        per_arm=per_arm, counters=objective_counters(grid_name),
        counters_source='profiles/%s[%s] (tools/perf/obj_counters.sh, '
                        'stand-alone objective bench; not this run)' % (
                            os.path.basename(profile_file('obj_counters.json')),
                            grid_name),
        note='flops / bytes: objective_model() of bench.py (phase by phase, DESIGN '
             '4.7); evaluations counted by the optimiser; seconds = the whole '
             'vel_fit.process call (first grid, Nelder-Mead, refinement, Hessian)')


def run_process_addon(batch, rec, arms, args, dev):
    """vel_fit.process on the first n spectra of the batch,
    started from the CCF parameters as desi_fit.py:289-309 does; the oracle's
    process (scipy Nelder-Mead) on a small sample beside it."""
    import torch
    from rvspecfit_amd import engine, pipeline, vel_fit
    n = min(args.process, batch.S)
    idx = torch.arange(n, device=dev)
    if any(a.G > 1 for a in batch.arms):   # (--workload sdss: a grid set)
        sub = batch.subset(idx)
    else:
        sub = engine.SpecBatch([engine.ArmData(a.name, a.lam_host, a.spec[idx],
                                               a.espec[idx], a.badmask[idx],
                                               device=dev) for a in batch.arms])
    F = pipeline.RECORD_FIELDS
    names = ['teff', 'logg', 'feh', 'alpha']
    r0 = rec[:n]
    pd0 = {k: r0[:, F.index('p%d' % i)].contiguous() for i, k in enumerate(names)}
    vs = r0[:, F.index('vsini')]
    pd0['vsini'] = torch.where(torch.isfinite(vs), vs,
                               torch.zeros_like(vs)).contiguous()
    cfg = dict(CONFIG)
    cfg.setdefault('max_vsini', 500)
    cfg['second_minimizer'] = (not args.process_no_bfgs)
    vel_fit.process(sub, pd0, options=OPTIONS, config=cfg)   # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = vel_fit.process(sub, pd0, options=OPTIONS, config=cfg)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the stage breakdown comes from a second, single-stream run (stage timers
    # switch off the two-halves split of vel_fit.process)
    tm = {}
    t1 = time.perf_counter()
    vel_fit.process(sub, pd0, options=OPTIONS, config=cfg, timers=tm)
    torch.cuda.synchronize()
    dt1 = time.perf_counter() - t1
    out = dict(spectra=n, spectra_per_s=round(n / dt, 1), seconds=round(dt, 2),
               streams=vel_fit.PROCESS_STREAMS if n >= vel_fit.PROCESS_SPLIT_MIN
               else 1,
               single_stream_seconds=round(dt1, 2),
               stage_s={k: round(v, 3) for k, v in tm.items()},
               nm_rounds=int(r['nm_rounds']),
               nm_iterations_mean=round(float(r['nm_nit'].float().mean()), 1),
               nm_iterations_max=int(r['nm_nit'].max()),
               objective_evals=int(r['objective_evals']),
               nm_launched_rows=int(r.get('nm_launched_rows', 0)),
               minimize_success=round(float(
                   r['minimize_success'].float().mean()), 4),
               bad_hessian=round(float(np.mean(r['bad_hessian'])), 4),
               second_minimizer=(not args.process_no_bfgs),
               note='add-on, not part of `value`; Hessian by central '
                    'differences (see DESIGN.md)')
    out['roofline'] = process_roofline(sub, r, out, tm, dt, dt1)
    if 'bfgs' in r:
        out['bfgs'] = dict(rounds=int(r['bfgs']['rounds']),
                           nfev_mean=round(float(np.mean(r['bfgs']['nfev'])), 1),
                           nit_mean=round(float(np.mean(r['bfgs']['nit'])), 2),
                           status_counts=np.bincount(
                               r['bfgs']['status'], minlength=4).tolist())
    m = min(args.process_cpu_sample, n)
    if m > 0 and not args.no_cpu_baseline:
        start = torch.stack([pd0[k] for k in names] + [pd0['vsini']],
                            dim=1).cpu().numpy()
        cb = run_cpu_baseline(arms, m, args, start=start)
        o = np.array(cb['recs'])
        gv = r['vel'][:m].cpu().numpy()
        gc = r['chisq'][:m].cpu().numpy()
        gp = np.stack([r['param'][k][:m].cpu().numpy() for k in names], axis=1)
        perr = o[:, 8:12]
        out['cpu'] = dict(value=round(cb['n'] / cb['wall'], 3),
                          unit='spectra/s', cores=cb['cores'], kind='port',
                          one_spectrum_seconds_1core=round(
                              cb['one_spectrum_s'], 2),
                          sample='%d spectra, oracle process (scipy Nelder-Mead '
                                 'on the C port of get_chisq)' % m)
        with np.errstate(all='ignore'):
            out['parity'] = dict(
                n=m, cpu_raised=int(np.isnan(o[:, 0]).sum()),
                max_abs_dvel=float(np.nanmax(np.abs(gv - o[:, 0]))),
                max_abs_dchisq=float(np.nanmax(np.abs(gc - o[:, 2]))),
                max_dparam_over_sigma=float(np.nanmax(
                    np.abs(gp - o[:, 4:8]) / np.where(perr > 0, perr, np.nan))))
    return out


def run_desi_addon(arms, args, dev, dicts):
    """the survey driver end to end on one synthetic coadd
    file of N fibres x 3 DESI arms (float32 flux/ivar, int32 mask, as the real
    files), from the FITS bytes on disk to the RVTAB/RVMOD products."""
    import tempfile
    import torch
    from rvspecfit_amd import fits_min as F
    from rvspecfit_amd.desi import desi_fit as D
    n = min(args.desi_file, arms[0][2].shape[0])
    tmp = tempfile.mkdtemp(prefix='rvs_desi_')
    fname = os.path.join(tmp, 'coadd-bench.fits')
    hdus = [F.PrimaryHDU()]
    hdus[0].header['SPGRP'] = 'healpix'
    fm = F.FitsTable()
    fm.add('TARGETID', np.arange(n, dtype=np.int64) + 39628000000000000)
    fm.add('FIBER', np.arange(n, dtype=np.int32))
    fm.add('TARGET_RA', np.linspace(150., 151., n))
    fm.add('TARGET_DEC', np.linspace(2., 3., n))
    fm.add('OBJTYPE', np.array(['TGT'] * n))
    fm.add('COADD_FIBERSTATUS', np.zeros(n, dtype=np.int32))
    fm.add('BRICKID', np.zeros(n, dtype=np.int32))
    hdus.append(F.BinTableHDU(fm, name='FIBERMAP'))
    sc = F.FitsTable()
    sc.add('TARGETID', fm['TARGETID'])
    nbytes = 0
    for name, lam, spec, es, bad in arms:
        A = name[-1].upper()
        flux = spec[:n].float().cpu().numpy()
        ivar = (1.0 / es[:n]**2).float().cpu().numpy()
        mask = bad[:n].to(torch.int32).cpu().numpy()
        hdus += [F.ImageHDU(np.asarray(lam, dtype=np.float64),
                            name=A + '_WAVELENGTH'),
                 F.ImageHDU(flux, name=A + '_FLUX'),
                 F.ImageHDU(ivar, name=A + '_IVAR'),
                 F.ImageHDU(mask, name=A + '_MASK')]
        sc.add('MEDIAN_COADD_SNR_' + A, D.get_sns(flux, ivar, mask).astype(
            np.float64))
        nbytes += flux.nbytes + ivar.nbytes + mask.nbytes
    hdus.append(F.BinTableHDU(sc, name='SCORES'))
    F.HDUList(hdus).writeto(fname)
    cfg = dict(CONFIG, second_minimizer=(not args.process_no_bfgs),
               config_file_path='synthetic')
    tabf, modf = os.path.join(tmp, 'rvtab.fits'), os.path.join(tmp, 'rvmod.fits')
    logging_off()
    D.proc_desi(fname, tabf, modf, None, cfg, doplot=False, minsn=-1e9,
                npoly=OPTIONS['npoly'], device=dev)      # warm-up
    tm = {}
    D.FIT_TIMES.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nfiles = max(1, args.desi_nfiles)
    if nfiles == 1:
        nfit = D.proc_desi(fname, tabf, modf, None, cfg, doplot=False,
                           minsn=-1e9, npoly=OPTIONS['npoly'], device=dev,
                           timers=tm)
    else:
        # the driver's file loop: groups of 4 files fitted together, the next
        # group read and conditioned by a worker thread meanwhile
        import yaml
        cfgy = {k: v for k, v in cfg.items() if k != 'config_file_path'}
        if args.desi_workers > 1:
            # worker processes load the libraries from disk (untimed)
            tl = os.path.join(tmp, 'templ')
            os.makedirs(tl)
            for name, d in dicts.items():
                np.savez(os.path.join(tl, 'rvsgpu_%s.npz' % name), **d)
            cfgy['template_lib'] = tl + '/'
        cfgf = os.path.join(tmp, 'config.yaml')
        with open(cfgf, 'w') as fp:
            yaml.safe_dump(cfgy, fp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        links = [fname]
        for i in range(1, nfiles):
            links.append(os.path.join(tmp, 'coadd-bench%d.fits' % i))
            os.symlink(fname, links[-1])
        st = os.path.join(tmp, 'status')
        D.proc_many(links, tmp, 'rvtab', 'rvmod', config_fname=cfgf, minsn=-1e9,
                    doplot=False, subdirs=False, npoly=OPTIONS['npoly'],
                    process_status_file=st, shard=(0, 1),
                    files_per_batch=min(max(1, args.desi_files_per_batch), nfiles),
                    nthreads=max(1, args.desi_workers))
        rows = [l.split() for l in open(st).read().strip().split('\n')]
        assert all(r[1] == 'SUCCESS' for r in rows), rows
        nfit = sum(int(r[2]) for r in rows)
        tabf = os.path.join(tmp, 'rvtab_coadd-bench.fits')
        modf = os.path.join(tmp, 'rvmod_coadd-bench.fits')
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tab = F.open(tabf, verify_checksum=True)['RVTAB'].data
    warn = np.asarray(tab['RVS_WARN'])
    out = dict(fibres=int(nfit), files=nfiles,
               files_per_batch=min(max(1, args.desi_files_per_batch), nfiles),
               workers=max(1, args.desi_workers) if nfiles > 1 else 1,
               fibres_per_s=round(nfit / dt, 1), seconds=round(dt, 2),
               stage_s={k: round(v, 3) for k, v in tm.items()} if nfiles == 1
               else {k: round(v, 3) for k, v in D.GROUP_TIMES.items()},
               input_MB=round(os.path.getsize(fname) / 1e6, 1),
               output_MB=round((os.path.getsize(tabf)
                                + os.path.getsize(modf)) / 1e6, 1),
               success_frac=round(float((warn == 0).mean()), 4),
               fit_stage_s={k: round(v, 3) for k, v in D.FIT_TIMES.items()} or None,
               second_minimizer=(not args.process_no_bfgs),
               note='add-on, not part of `value`: desi_fit.proc_desi on one '
                    'synthetic coadd file, FITS in -> RVTAB/RVMOD out')
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return out


def logging_off():
    import logging
    logging.getLogger().setLevel(logging.ERROR)


def stage_round(d):
    return {k: round(v, 2) for k, v in d.items()}


if __name__ == '__main__':
    main()
