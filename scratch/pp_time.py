import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
from rvspecfit_amd import engine, spec_inter, _lib
from rvspecfit_amd.library import TemplateLibrary
dev = torch.device('cuda')
def gpu_convolve(lam, templ, vsini):
    t = torch.as_tensor(np.ascontiguousarray(templ)).to(dev); v = torch.as_tensor(np.ascontiguousarray(vsini)).to(dev)
    return engine.convolve_vsini(lam, t, v).cpu().numpy()
dicts = bench.build_library_dicts(64, gpu_convolve)
libs = {n: TemplateLibrary(n, d, device=dev) for n, d in dicts.items()}
S = 2000
arms = bench.make_spectra_device(bench.truth_params(S, 3), dev)
batch = engine.SpecBatch([engine.ArmData(n, lam, sp, es, bad, device=dev) for n, lam, sp, es, bad in arms])
for cont in (True, False):
    for n in libs: libs[n].ccf['continuum'] = cont
    for a in batch.arms: a._ccf.clear()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        for a in batch.arms:
            r = engine.ccf_preprocess(a, libs[a.name], bench.CONFIG)
        torch.cuda.synchronize(); dt = time.time() - t0
    print('continuum', cont, 'preprocess ms', dt * 1e3)
