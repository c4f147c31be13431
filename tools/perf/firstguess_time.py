import sys, time, numpy as np, torch
sys.path.insert(0,'.')
import bench
from rvspecfit_amd import engine, spec_inter, vel_fit
from rvspecfit_amd.library import TemplateLibrary
S=int(sys.argv[1]); dev=torch.device('cuda',0)
def conv(lam, templ, vsini):
    t=torch.as_tensor(np.ascontiguousarray(templ)).to(dev); v=torch.as_tensor(np.ascontiguousarray(vsini)).to(dev)
    return engine.convolve_vsini(lam,t,v).cpu().numpy()
for name,d in bench.build_library_dicts(64,conv).items():
    spec_inter.register_library(TemplateLibrary(name,d,device=dev), bench.CONFIG['template_lib'])
arms=bench.make_spectra_device(bench.truth_params(S,seed=3),dev)
batch=engine.SpecBatch([engine.ArmData(n,lam,sp,es,bad,device=dev) for n,lam,sp,es,bad in arms])
for it in range(2):
    torch.cuda.synchronize(); t0=time.time()
    g=vel_fit.firstguess(batch, options=bench.OPTIONS, config=bench.CONFIG)
    torch.cuda.synchronize(); print('firstguess S=%d: %.2f s (%.1f spectra/s)'%(S,time.time()-t0,S/(time.time()-t0)))
