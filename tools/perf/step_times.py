"""per-step wall time of the default bench workload in ONE process (cold-start
study): python tools/perf/step_times.py [nsteps]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
from rvspecfit_amd import engine, pipeline, spec_inter
from rvspecfit_amd.library import TemplateLibrary
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda', 0)
def gpu_convolve(lam, templ, vsini):
    t = torch.as_tensor(np.ascontiguousarray(templ)).to(dev)
    v = torch.as_tensor(np.ascontiguousarray(vsini)).to(dev)
    return engine.convolve_vsini(lam, t, v).cpu().numpy()
dicts = bench.build_library_dicts(64, gpu_convolve)
for name, d in dicts.items():
    spec_inter.register_library(TemplateLibrary(name, d, device=dev), bench.CONFIG['template_lib'])
tp = bench.truth_params(10000, seed=3)
arms = bench.make_spectra_device(tp, dev)
batch = engine.SpecBatch([engine.ArmData(nm, lam, sp, es, bad, device=dev) for nm, lam, sp, es, bad in arms])
torch.cuda.synchronize()
out = []
for i in range(n):
    for a in batch.arms:
        a._work.clear()
    tm = {}
    t0 = time.perf_counter()
    rec = pipeline.fit_batch(batch, bench.CONFIG, options=bench.OPTIONS, timers=tm)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    st = {k: round(v[0].elapsed_time(v[1]), 1) for k, v in tm.items()}
    out.append((round(dt, 1), st))
for o in out:
    print(o)
