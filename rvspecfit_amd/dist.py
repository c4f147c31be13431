"""Multi-GPU: spectra are independent units, so the path shards by spectrum
index with NO data-path collective; the only exchange is one gather of the
fixed-size per-spectrum result records.

One process per GPU (torchrun / torch.distributed.run); backend "nccl" is RCCL
over xGMI on ROCm.  The reference's counterpart is its process pool over
spectra (desi/desi_fit.py:1215-1218, 1475-1479).
"""
import os

import torch
import torch.distributed as dist


def shard_range(S, rank, world):
    """Contiguous index block [lo, hi) of rank `rank`: ceil(S/world) spectra per
    rank, the last ranks may be short or empty."""
    per = -(-S // world)
    lo = min(S, rank * per)
    hi = min(S, lo + per)
    return lo, hi


class DistInitError(RuntimeError):
    """the process group could not be formed (or its first collective failed)"""


def init_from_env(backend=None, probe=True, timeout_s=600):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (as set by
    torch.distributed.run).  Returns (rank, world, local_rank).

    RCCL builds its communicator lazily, on the first collective; with `probe` a
    one-element all_reduce runs right here so that a broken fabric / IPC set-up
    fails at start-up, not after the first step.  Any failure raises
    DistInitError once: there is NO retry and no fallback to another backend --
    this process may already have initialised the GPU, and such a process must
    neither be re-executed nor silently measure something else; the launcher
    (bench.py, torch.distributed.run) sees the non-zero exit and ends the job."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        import datetime
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = os.environ.get('RVS_DIST_BACKEND') or backend
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        try:
            dist.init_process_group(
                backend, timeout=datetime.timedelta(seconds=timeout_s))
            if probe:
                dev = 'cuda' if backend == 'nccl' else 'cpu'
                one = torch.ones(1, dtype=torch.float64, device=dev)
                dist.all_reduce(one)
                if dev == 'cuda':
                    torch.cuda.synchronize()
                if float(one.item()) != float(world):
                    raise DistInitError('probe all_reduce returned %r for a world '
                                        'of %d' % (float(one.item()), world))
        except DistInitError:
            raise
        except Exception as e:  # noqa: BLE001
            raise DistInitError(
                'rank %d of %d: process group (%s) failed at start-up: %s: %s'
                % (rank, world, backend, type(e).__name__, e)) from e
    return rank, world, local


def gather_records(rec, S_total, rank=None, world=None, alone_too=False):
    """All-gather per-shard record tensors [n_r, NREC] into [S_total, NREC] on
    every rank (shards follow shard_range; short shards are zero padded for
    the fixed-size collective and trimmed afterwards).  A group of one returns its
    records as they are; `alone_too` runs the collective even then (what a
    one-GPU box can show of the RCCL path: tests/test_dist_gpu.py)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1 and not (alone_too and dist.is_initialized()):
        return rec
    per = -(-S_total // world)
    nrec = rec.shape[1]
    pad = torch.zeros((per, nrec), dtype=rec.dtype, device=rec.device)
    pad[:rec.shape[0]] = rec
    full = torch.empty((world * per, nrec), dtype=rec.dtype, device=rec.device)
    if rec.is_cuda and dist.get_backend() == 'nccl':
        dist.all_gather_into_tensor(full, pad)   # RCCL over xGMI
    else:  # gloo (CPU tests, or a functional multi-rank run without RCCL)
        hpad = pad.cpu()
        parts = [torch.empty_like(hpad) for _ in range(world)]
        dist.all_gather(parts, hpad)
        full = torch.cat(parts, dim=0).to(rec.device)
    return full[:S_total]


def fit_sharded(make_batch, S_total, config, options=None, refine=False,
                process=False):
    """Every rank fits its index block and all ranks receive the full table.

    make_batch(lo, hi) -> engine.SpecBatch holding spectra lo..hi-1 on this
    rank's GPU (the template libraries are replicated per rank).  process=True
    appends the optimiser stage (pipeline.process_batch): the gathered record is
    then [S_total, NREC + NPROC]."""
    from . import pipeline
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_range(S_total, rank, world)
    width = pipeline.NREC + (pipeline.NPROC if process else 0)
    if hi > lo:
        batch = make_batch(lo, hi)
        rec = pipeline.fit_batch(batch, config, options=options, refine=refine)
        if process:
            rec = torch.cat([rec, pipeline.process_batch(batch, rec, config,
                                                         options=options)],
                            dim=1)
    else:
        dev = 'cuda' if torch.cuda.is_available() else 'cpu'
        rec = torch.zeros((0, width), dtype=torch.float64, device=dev)
    return gather_records(rec, S_total, rank, world)
