"""cProfile of the Hessian stage of vel_fit.process (vel_fit.py:699-725) on S bench
spectra: where its wall time goes (objective launches vs host post-processing).
usage: python tools/perf/hessian_profile.py [S]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from rvspecfit_amd import engine, pipeline, spec_inter, vel_fit  # noqa: E402
from rvspecfit_amd.library import TemplateLibrary  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dev = torch.device('cuda', 0)


def conv(lam, templ, vsini):
    t = torch.as_tensor(np.ascontiguousarray(templ)).to(dev)
    v = torch.as_tensor(np.ascontiguousarray(vsini)).to(dev)
    return engine.convolve_vsini(lam, t, v).cpu().numpy()


for name, d in bench.build_library_dicts(64, conv).items():
    spec_inter.register_library(TemplateLibrary(name, d, device=dev),
                                bench.CONFIG['template_lib'])
arms = bench.make_spectra_device(bench.truth_params(S, seed=3), dev)
batch = engine.SpecBatch([engine.ArmData(n, lam, sp, es, bad, device=dev)
                          for n, lam, sp, es, bad in arms])
rec = pipeline.fit_batch(batch, bench.CONFIG, options=bench.OPTIONS)
names = ['teff', 'logg', 'feh', 'alpha']
params = rec[:, 2:6].contiguous()
vs = torch.where(torch.isfinite(rec[:, 6]), rec[:, 6], torch.zeros_like(rec[:, 6]))
vel = rec[:, 7].contiguous()
import types  # noqa: E402
cfg = dict(bench.CONFIG, max_vsini=500)
obj = vel_fit._Objective(batch, types.SimpleNamespace(specParams=names), cfg,
                         bench.OPTIONS, None)
vel_fit._hessian_stage(obj, names, vel, params, vs)      # warm-up
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
err, cov, bad = vel_fit._hessian_stage(obj, names, vel, params, vs)
torch.cuda.synchronize()
pr.disable()
print('hessian stage %.3f s for %d spectra; bad %.3f' %
      (time.perf_counter() - t0, S, bad.mean()))
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
