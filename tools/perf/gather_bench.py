"""rvs_template_polylinear alone, at S random in-grid parameter vectors, on a
synthetic library of a given grid size (one DESI arm): time per launch and
algorithmic GB/s.  Under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` (gather_pmc.sh)
every polylinear_kernel launch of the process is such a launch, so the counters give
the real HBM bytes of the gather.
usage: python tools/perf/gather_bench.py [nteff,nlogg,nfeh,nalpha] [S] [arm]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from rvspecfit_amd import synth  # noqa: E402
from rvspecfit_amd.library import TemplateLibrary  # noqa: E402

g = [int(_) for _ in (sys.argv[1] if len(sys.argv) > 1 else '40,11,8,5').split(',')]
S = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
arm = sys.argv[3] if len(sys.argv) > 3 else 'b'
dev = torch.device('cuda', 0)
l0, l1, st = synth.DESI_ARMS[arm]['templ']
lib = synth.make_interp_library_fast('desi_' + arm, l0, l1, st, grid_kw=dict(
    nteff=g[0], nlogg=g[1], nfeh=g[2], nalpha=g[3]), resol=3000., device=dev)
L = TemplateLibrary('desi_' + arm, synth.library_as_npz_dict(lib), device=dev)
rng = np.random.RandomState(3)
P = torch.as_tensor(np.stack([rng.uniform(3500, 11500, S), rng.uniform(0.3, 4.7, S),
                              rng.uniform(-1.9, -0.1, S),
                              rng.uniform(0.05, 0.95, S)], axis=1)).to(dev)
L.eval_batch(P)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 5
e0.record()
for _ in range(n):
    L.eval_batch(P)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
alg = S * L.ntp * (16 * 4 + 8)
print('grid %s (%d templates, %.0f MB) S %d: %.3f ms per launch, algorithmic '
      '%.1f GB/s (%.2f GB per launch: 16 float32 rows in + float64 template out)'
      % ('x'.join(map(str, g)), L.ngrid, L.ngrid * L.ntp * 4 / 1e6, S, ms,
         alg / ms / 1e6, alg / 1e9))
