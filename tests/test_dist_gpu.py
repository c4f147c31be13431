"""Multi-rank path on hardware without a multi-GPU node: `bench.py --gpus 2` with
two gloo ranks sharing ONE GPU (RVS_SHARE_GPU=1 RVS_DIST_BACKEND=gloo) -- the
launch path, the per-rank shards and the record gather -- against two
single-rank runs of the same shards, row for row; and BASELINE configs[4]'s
per-GPU shard (62 500 spectra) through size-independent properties and an
oracle sample.  The RCCL collective itself needs a multi-GPU node (the driver's
scaling run); SCALE_r*.json says whether one was available."""
import argparse
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, 'bench.py')


def _run(argv, **env):
    e = {k: v for k, v in os.environ.items()
         if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR',
                      'MASTER_PORT')}
    e.update(env)
    out = subprocess.run([sys.executable, BENCH] + argv, env=e, text=True,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines()
                       if l.lstrip().startswith('{')][-1])


def test_two_ranks_on_one_gpu_equal_two_single_rank_runs(tmp_path):
    S = 300
    common = ['--spectra', str(S), '--steps', '1', '--warmup', '0',
              '--no-cpu-baseline']
    f2 = str(tmp_path / 'two.npy')
    line = _run(['--gpus', '2', '--dump-records', f2] + common,
                RVS_SHARE_GPU='1', RVS_DIST_BACKEND='gloo')
    assert line['n_gpus'] == 2
    assert line['config']['parallelism'] == 'spectra-sharded x2'
    two = np.load(f2)
    assert two.shape == (2 * S, 16)
    for r in range(2):
        f1 = str(tmp_path / ('one%d.npy' % r))
        one = _run(['--gpus', '1', '--seed-rank', str(r), '--dump-records', f1]
                   + common)
        assert one['n_gpus'] == 1
        # rank r's shard sits at rows [r S, (r+1) S) of the gathered table
        np.testing.assert_array_equal(two[r * S:(r + 1) * S], np.load(f1))
    # the shards are different spectra
    assert not np.array_equal(two[:S], two[S:])


def test_rccl_group_of_one_runs_the_gather(tmp_path):
    """what ONE GPU can show of the RCCL branch: a process group of one rank on
    backend "nccl" (= RCCL) forms, the start-up probe's all_reduce and
    dist.gather_records' all_gather_into_tensor run on the device and return the
    shard's records unchanged (short shard: padded, gathered, trimmed).  The
    communicator over xGMI between GPUs is the driver's scaling run."""
    code = r'''
import os, sys, socket, torch
sys.path.insert(0, %r)
s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                  MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
import torch.distributed as td
from rvspecfit_amd import dist
td.init_process_group('nccl', rank=0, world_size=1)
assert td.get_backend() == 'nccl'
one = torch.ones(1, dtype=torch.float64, device='cuda')
td.all_reduce(one); torch.cuda.synchronize()
assert float(one.item()) == 1.0
rec = torch.arange(37 * 16, dtype=torch.float64, device='cuda').reshape(37, 16)
out = dist.gather_records(rec, 37, alone_too=True)
torch.cuda.synchronize()
assert out.data_ptr() != rec.data_ptr() and torch.equal(out, rec)
td.destroy_process_group()
print('RCCL_ONE_RANK_OK')
''' % REPO
    e = {k: v for k, v in os.environ.items()
         if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR',
                      'MASTER_PORT')}
    e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    out = subprocess.run([sys.executable, '-c', code], env=e, text=True,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=600)
    assert out.returncode == 0 and 'RCCL_ONE_RANK_OK' in out.stdout, \
        out.stderr[-3000:]


def test_config4_shard_62500_spectra():
    """one GPU's share of BASELINE configs[4] (500 000 spectra over 8 GPUs): the
    record of a spectrum does not depend on the 62 499 others (subsets fitted
    alone: bit for bit), and a sample agrees with the CPU oracle"""
    import bench
    from rvspecfit_amd import _lib, engine, pipeline, spec_inter
    from rvspecfit_amd.library import TemplateLibrary
    _lib.require_gpu()
    dev = torch.device('cuda', 0)
    S = 62500

    def gpu_convolve(lam, templ, vsini):
        t = torch.as_tensor(np.ascontiguousarray(templ)).to(dev)
        v = torch.as_tensor(np.ascontiguousarray(vsini)).to(dev)
        return engine.convolve_vsini(lam, t, v).cpu().numpy()
    for name, d in bench.build_library_dicts(64, gpu_convolve).items():
        spec_inter.register_library(TemplateLibrary(name, d, device=dev),
                                    bench.CONFIG['template_lib'])
    tp = bench.truth_params(S, seed=3 + 1000 * 5)          # rank 5's shard
    sample = [0, 31249, 62499, 40000]
    tp['snr'][sample] = [50., 300., 1000., 100.]
    arms = bench.make_spectra_device(tp, dev)
    batch = engine.SpecBatch([engine.ArmData(n, lam, sp, es, bad, device=dev)
                              for n, lam, sp, es, bad in arms])
    rec = pipeline.fit_batch(batch, bench.CONFIG, options=bench.OPTIONS)
    F = pipeline.RECORD_FIELDS
    r = rec.cpu().numpy()
    assert r.shape == (S, pipeline.NREC)
    assert np.isfinite(r[:, F.index('best_vel')]).all()
    assert (r[:, F.index('status')] == 0).mean() > 0.99
    assert np.median(np.abs(r[:, F.index('best_vel')] - tp['vel'])) < 3.0
    # subsets alone: the ends, a stride across the CCF accumulator chunks
    for ix in (torch.arange(0, 65, device=dev),
               torch.arange(S - 200, S, device=dev),
               torch.arange(7, S, 997, device=dev)):
        sub = pipeline.fit_batch(batch.subset(ix), bench.CONFIG,
                                 options=bench.OPTIONS)
        assert np.array_equal(sub.cpu().numpy(), rec[ix].cpu().numpy(),
                              equal_nan=True)
    # oracle sample (north-star tolerances; chi^2 relative to max(|chi|, npix))
    ix = torch.as_tensor(sample)
    sarms = [(nm, lam, sp[ix.to(dev)], es[ix.to(dev)], bad[ix.to(dev)])
             for nm, lam, sp, es, bad in arms]
    args = argparse.Namespace(ccf_every=64, cpu_cores=4, workload='desi',
                              evaluator='polylinear', grid='')
    o = np.array(bench.run_cpu_baseline(sarms, len(sample), args)['recs'])
    g = r[sample]
    assert np.array_equal(g[:, F.index('best_id')], o[:, 0])
    assert np.abs(g[:, F.index('vrad_ccf')] - o[:, 1]).max() < 1e-2
    assert np.abs(g[:, F.index('best_vel')] - o[:, 2]).max() < 1e-3
    npix_tot = sum(a[2].shape[1] for a in arms)
    rel = np.abs(g[:, F.index('best_chi')] - o[:, 4]) / \
        np.maximum(np.abs(o[:, 4]), npix_tot)
    assert rel.max() < 1e-9, rel
