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
