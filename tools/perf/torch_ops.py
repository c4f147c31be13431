"""Which torch (non-HIP-extension) kernels one bench step launches, by call site.
   python tools/perf/torch_ops.py  -> table of aten ops with counts and the
   rvspecfit_amd source line that issued them."""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.argv = ['bench.py', '--spectra', '2000', '--steps', '1', '--warmup', '1',
            '--no-cpu-baseline']
import bench  # noqa: E402

counts = collections.Counter()


class Mode(torch.utils._python_dispatch.TorchDispatchMode):

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        dev = any(isinstance(a, torch.Tensor) and a.is_cuda
                  for a in list(args) + list((kwargs or {}).values()))
        if dev or (isinstance(out, torch.Tensor) and out.is_cuda):
            site = '?'
            for fr in reversed(traceback.extract_stack()):
                if 'rvspecfit_amd' in fr.filename and 'torch' not in fr.filename:
                    site = '%s:%d' % (os.path.basename(fr.filename), fr.lineno)
                    break
            counts[(site, str(func))] += 1
        return out


from rvspecfit_amd import pipeline  # noqa: E402
orig = pipeline.fit_batch
state = {'n': 0}


def wrapped(*a, **k):
    state['n'] += 1
    if state['n'] == 3:   # warm-up, timed step, then this one
        with Mode():
            return orig(*a, **k)
    return orig(*a, **k)


pipeline.fit_batch = wrapped
bench.main()
by_site = collections.Counter()
for (site, f), n in counts.items():
    by_site[site] += n
print('total dispatched ops on device tensors:', sum(counts.values()))
for site, n in by_site.most_common(45):
    ops = ', '.join('%s x%d' % (f.replace('aten.', ''), c) for (s, f), c in
                    sorted(counts.items(), key=lambda kv: -kv[1]) if s == site)
    print('%5d  %-24s %s' % (n, site, ops[:150]))
