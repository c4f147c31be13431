"""vel_fit.process on Delaunay libraries: the rounds inside the library against the
rounds driven from Python (optimizer.NATIVE_ROUNDS), stage times and counters.
   python tools/perf/tri_rounds_ab.py [S=300]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
from rvspecfit_amd import engine, optimizer, pipeline, spec_inter, vel_fit  # noqa: E402
from rvspecfit_amd.library import TemplateLibrary  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bench.EVALUATOR = sys.argv[2] if len(sys.argv) > 2 else 'tri'
dev = torch.device('cuda', 0)


def conv(lam, templ, vsini):
    t = torch.as_tensor(np.ascontiguousarray(templ)).to(dev)
    v = torch.as_tensor(np.ascontiguousarray(vsini)).to(dev)
    return engine.convolve_vsini(lam, t, v).cpu().numpy()


dicts = bench.build_library_dicts(64, conv)
for name, d in dicts.items():
    spec_inter.register_library(TemplateLibrary(name, d, device=dev),
                                bench.CONFIG['template_lib'])
tp = bench.truth_params(S, seed=3)
arms = bench.make_spectra_device(tp, dev)
batch = engine.SpecBatch([engine.ArmData(n, lam, sp, es, bad, device=dev)
                          for n, lam, sp, es, bad in arms])
rec = pipeline.fit_batch(batch, bench.CONFIG, options=bench.OPTIONS)
F = pipeline.RECORD_FIELDS
names = ['teff', 'logg', 'feh', 'alpha']
pd0 = {k: rec[:, F.index('p%d' % i)].contiguous() for i, k in enumerate(names)}
vs = rec[:, F.index('vsini')]
pd0['vsini'] = torch.where(torch.isfinite(vs), vs, torch.zeros_like(vs)).contiguous()
cfg = dict(bench.CONFIG, max_vsini=500, second_minimizer=False)
for native in (True, False, True):
    optimizer.NATIVE_ROUNDS = native
    tm = {}
    torch.cuda.synchronize()
    t0 = time.time()
    r = vel_fit.process(batch, pd0, options=bench.OPTIONS, config=cfg, timers=tm)
    torch.cuda.synchronize()
    print(json.dumps(dict(native=native, seconds=round(time.time() - t0, 3),
                          stage={k: round(v, 3) for k, v in tm.items()},
                          rounds=int(r['nm_rounds']), evals=int(r['objective_evals']),
                          slots=int(r['nm_launched_rows']),
                          nit_max=int(r['nm_nit'].max()))), flush=True)
