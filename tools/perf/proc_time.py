import os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
from rvspecfit_amd import _lib, engine, pipeline, spec_inter, vel_fit
from rvspecfit_amd.library import TemplateLibrary
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device('cuda', 0)
def gpu_convolve(lam, templ, vsini):
    t = torch.as_tensor(np.ascontiguousarray(templ)).to(dev)
    v = torch.as_tensor(np.ascontiguousarray(vsini)).to(dev)
    return engine.convolve_vsini(lam, t, v).cpu().numpy()
dicts = bench.build_library_dicts(64, gpu_convolve)
for name, d in dicts.items():
    spec_inter.register_library(TemplateLibrary(name, d, device=dev), bench.CONFIG['template_lib'])
tp = bench.truth_params(S, seed=3)
arms = bench.make_spectra_device(tp, dev)
batch = engine.SpecBatch([engine.ArmData(n, lam, sp, es, bad, device=dev) for n, lam, sp, es, bad in arms])
rec = pipeline.fit_batch(batch, bench.CONFIG, options=bench.OPTIONS)
F = pipeline.RECORD_FIELDS
names = ['teff', 'logg', 'feh', 'alpha']
pd0 = {k: rec[:, F.index('p%d' % i)].contiguous() for i, k in enumerate(names)}
vs = rec[:, F.index('vsini')]
pd0['vsini'] = torch.where(torch.isfinite(vs), vs, torch.zeros_like(vs)).contiguous()
cfg = dict(bench.CONFIG)
for it in range(2):
    tm = {}
    torch.cuda.synchronize(); t0 = time.time()
    r = vel_fit.process(batch, pd0, options=bench.OPTIONS, config=cfg, timers=tm)
    torch.cuda.synchronize(); dt = time.time() - t0
    print('S', S, 'time %.2f' % dt, 'spectra/s %.1f' % (S / dt), {k: round(v, 2) for k, v in tm.items()},
          'rounds', r['nm_rounds'], 'evals', r['objective_evals'], 'nit mean', float(r['nm_nit'].float().mean()), 'max', int(r['nm_nit'].max()),
          'success', float(r['minimize_success'].float().mean()), 'bad_hess', r['bad_hessian'].mean())
tv = torch.as_tensor(tp['vel']).to(dev) if isinstance(tp, dict) and 'vel' in tp else None
print({k: v for k, v in tp.items()}.keys() if isinstance(tp, dict) else type(tp))
