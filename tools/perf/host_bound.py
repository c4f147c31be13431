"""Is a small step host bound?  Time to ENQUEUE one pipeline.fit_batch (no
synchronisation) against the time of the step itself:
python tools/perf/host_bound.py [workload cfg2|desi] [S]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
if wl == 'cfg2':
    bench.ARMS = ('c', )
from rvspecfit_amd import engine, pipeline, spec_inter
from rvspecfit_amd.library import TemplateLibrary
dev = torch.device('cuda', 0)


def gpu_convolve(lam, templ, vsini):
    t = torch.as_tensor(np.ascontiguousarray(templ)).to(dev)
    v = torch.as_tensor(np.ascontiguousarray(vsini)).to(dev)
    return engine.convolve_vsini(lam, t, v).cpu().numpy()


for name, d in bench.build_library_dicts(64, gpu_convolve).items():
    spec_inter.register_library(TemplateLibrary(name, d, device=dev),
                                bench.CONFIG['template_lib'])
arms = bench.make_spectra_device(bench.truth_params(S, seed=3), dev)
batch = engine.SpecBatch([engine.ArmData(n, lam, sp, es, bad, device=dev)
                          for n, lam, sp, es, bad in arms])


def step():
    for a in batch.arms:
        a._work.clear()
    return pipeline.fit_batch(batch, bench.CONFIG, options=bench.OPTIONS)


for _ in range(3):
    step()
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    step()
torch.cuda.synchronize()
full = (time.perf_counter() - t0) / N
# enqueue only: the deferred redo look is the one synchronisation of a step
real = engine.torch.nonzero
enq = []
for _ in range(N):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    engine.torch.nonzero = lambda x: torch.zeros((0, 1), dtype=torch.long, device=dev)
    try:
        step()
    finally:
        engine.torch.nonzero = real
    enq.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print('%s S %d: step %.2f ms, host time to enqueue it %.2f ms (median)' % (
    wl, S, full * 1e3, 1e3 * float(np.median(enq))))
