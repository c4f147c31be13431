"""process() as one batch vs two concurrent halves (vel_fit.PROCESS_STREAMS):
wall time and bit-equality.  python tools/perf/proc_split.py [nspectra] [bfgs]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
from rvspecfit_amd import engine, pipeline, spec_inter, vel_fit
from rvspecfit_amd.library import TemplateLibrary
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
bf = len(sys.argv) > 2
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
cfg = dict(bench.CONFIG, second_minimizer=bf)
out = {}
for mode in (1, 2, 1, 2, 1, 2, 2, 1, 1, 2):
    vel_fit.PROCESS_STREAMS = mode
    torch.cuda.synchronize(); t0 = time.time()
    r = vel_fit.process(batch, pd0, options=bench.OPTIONS, config=cfg)
    torch.cuda.synchronize(); dt = time.time() - t0
    print('streams', mode, 'time %.3f' % dt, 'spectra/s %.1f' % (S / dt))
    out[mode] = r
a, b = out[1], out[2]
for k in ('vel', 'vel_err', 'chisq', 'nm_nit', 'nm_nfev'):
    print(k, 'equal', bool(torch.equal(a[k], b[k])))
print('param equal', all(torch.equal(a['param'][k], b['param'][k]) for k in names), 'param_err equal', all(np.array_equal(a['param_err'][k], b['param_err'][k], equal_nan=True) for k in names), 'yfit equal', all(torch.equal(x, y) for x, y in zip(a['yfit'], b['yfit'])))
