"""N>1 path on CPU: world_size-2 gloo processes exercise the index sharding and
the single record gather (the only collective of the path)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rvspecfit_amd import dist as rdist

NREC = 16


def test_shard_range_covers_everything_once():
    for S in (0, 1, 7, 10, 1000, 1001):
        for W in (1, 2, 3, 8):
            seen = np.zeros(S, dtype=int)
            sizes = []
            for r in range(W):
                lo, hi = rdist.shard_range(S, r, W)
                assert 0 <= lo <= hi <= S
                seen[lo:hi] += 1
                sizes.append(hi - lo)
            assert np.all(seen == 1)
            assert max(sizes) <= -(-S // W) if S else True


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, S, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = rdist.init_from_env(backend='gloo')
    lo, hi = rdist.shard_range(S, r, w)
    idx = torch.arange(lo, hi, dtype=torch.float64)
    # a "record" that encodes its global index, as fit_batch would return
    rec = idx[:, None] * 100 + torch.arange(NREC, dtype=torch.float64)[None, :]
    full = rdist.gather_records(rec, S, r, w)
    q.put((rank, full.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('S', [10, 7, 1])
def test_gather_records_gloo_world2(S):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, S, q))
             for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    expect = (np.arange(S)[:, None] * 100 + np.arange(NREC)[None, :]).astype(float)
    for r in range(world):
        np.testing.assert_array_equal(got[r], expect)


# ---- bench.py --gpus N: the command the driver's scaling run uses -----------
import json
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, 'bench.py')


def _bench(argv, **env):
    e = {k: v for k, v in os.environ.items()
         if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR',
                      'MASTER_PORT')}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + argv, env=e, text=True,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=300)


def test_bench_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two fresh
    rank processes that find each other (gloo group formed from the variables
    the parent set) and relays exactly rank 0's line."""
    out = _bench(['--gpus', '2', '--dry-launch'])
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['WORLD_SIZE'] == '2' and d['RANK'] == '0'
    assert d['ranks'] == [0, 1]
    assert d['MASTER_ADDR'] == '127.0.0.1'
    assert len(set(d['pids'])) == 2 and os.getpid() not in d['pids']
    assert d['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_bench_fails_when_a_rank_fails():
    for bad in ('0', '1'):
        out = _bench(['--gpus', '2', '--dry-launch'], RVS_BENCH_FAIL_RANK=bad)
        assert out.returncode == 3, (bad, out.returncode, out.stderr)
        assert not out.stdout.strip()


def test_bench_refuses_gpus_different_from_world():
    # under a launcher: --gpus must be the world the launcher formed
    out = _bench(['--gpus', '2', '--dry-launch'], WORLD_SIZE='4', RANK='0')
    assert out.returncode == 2 and 'refusing' in out.stderr
    out = _bench(['--gpus', '1', '--dry-launch'], WORLD_SIZE='2', RANK='0')
    assert out.returncode == 2
    out = _bench(['--dry-launch'])
    assert out.returncode == 0 and json.loads(out.stdout)['n_gpus'] == 1


def test_bench_under_torch_distributed_run():
    """the driver's N>1 command line: torch.distributed.run starts the ranks,
    bench.py must then NOT start any of its own"""
    e = {k: v for k, v in os.environ.items()
         if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR',
                      'MASTER_PORT')}
    port = _free_port()
    out = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
         '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', str(port), BENCH, '--gpus', '2', '--dry-launch'],
        env=e, text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
        timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    js = [json.loads(l) for l in out.stdout.splitlines()
          if l.lstrip().startswith('{')]
    assert len(js) == 1 and js[0]['n_gpus'] == 2 and js[0]['ranks'] == [0, 1]


def test_init_failure_raises_once_and_leaves_no_group(monkeypatch):
    """a process group that cannot be formed raises DistInitError -- no retry,
    no fallback backend (the caller exits non-zero; bench.py: code 4)"""
    monkeypatch.setenv('WORLD_SIZE', '2')
    monkeypatch.setenv('RANK', '0')
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', str(_free_port()))
    calls = []
    real = dist.init_process_group

    def failing(*a, **k):
        calls.append(a)
        raise RuntimeError('simulated RCCL failure')
    monkeypatch.setattr(dist, 'init_process_group', failing)
    with pytest.raises(rdist.DistInitError) as e:
        rdist.init_from_env(backend='nccl')
    assert 'simulated RCCL failure' in str(e.value) and len(calls) == 1
    assert not dist.is_initialized()
    monkeypatch.setattr(dist, 'init_process_group', real)
