"""Experiment: does the CCF stage (FFT kernel, LDS/barrier-bound) of one part of a
batch overlap usefully with the chi^2 grid (fp64 VALU-bound) of another part?
Two host threads / two streams, parts alternate; a lock per stage keeps the two
threads in complementary stages.  python tools/perf/stage_overlap.py [S] [nparts]"""
import sys, time, threading, numpy as np, torch
sys.path.insert(0, '.')
import bench
from rvspecfit_amd import engine, pipeline, spec_inter
from rvspecfit_amd.library import TemplateLibrary
S = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 4
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
cfg, opt = bench.CONFIG, bench.OPTIONS
libs = spec_inter.get_libs(batch.names, cfg)
vg = torch.as_tensor(np.arange(cfg['min_vel'], cfg['max_vel'], cfg['vel_step0']).astype(np.float64)).to(dev)
npoly = opt.get('npoly') or 5

def stage_ccf(b):
    ccf = engine.ccf_fit(b, libs, cfg)
    ref = libs[b.names[0]].ccf
    params = ref['params_dev'][ccf['best_id']].contiguous()
    vsini = ref['vsinis_dev'][ccf['best_id']].contiguous()
    return ccf, params, vsini

def stage_chisq(b, params, vsini):
    coefs, outs = [], []
    for arm in b.arms:
        c, o = engine.build_templates(libs[arm.name], params, vsini)
        coefs.append(c); outs.append(o)
    chisq, status = engine.chisq_grid(b, libs, coefs, outs, vg, npoly=npoly, rbf=True,
                                      vel_bounds=(float(cfg['min_vel']), float(cfg['max_vel'])))
    res, _, mst = engine.grid_moments(chisq, vg, Np=1)
    cont = engine.chisq_continuum(b, npoly=npoly, rbf=True)
    return res, cont

def run_single():
    c, p, v = stage_ccf(batch)
    r, _ = stage_chisq(batch, p, v)
    torch.cuda.synchronize()
    return c['best_vel'], r

parts = [batch.subset(torch.arange(k, S, NP, device=dev)) for k in range(NP)]
torch.cuda.synchronize()
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
lock_ccf, lock_chi = threading.Lock(), threading.Lock()

def run_overlap():
    out = [None] * NP
    def worker(t):
        with torch.cuda.stream(streams[t]):
            for k in range(t, NP, 2):
                with lock_ccf:
                    c, p, v = stage_ccf(parts[k])
                    streams[t].synchronize()
                with lock_chi:
                    r, _ = stage_chisq(parts[k], p, v)
                    streams[t].synchronize()
                out[k] = (c['best_vel'], r)
    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return out

def run_parts_serial():
    out = []
    for k in range(NP):
        c, p, v = stage_ccf(parts[k])
        r, _ = stage_chisq(parts[k], p, v)
        out.append((c['best_vel'], r))
    torch.cuda.synchronize()
    return out

for name, fn in (('single', run_single), ('parts serial', run_parts_serial), ('overlap', run_overlap)) * 3:
    torch.cuda.synchronize(); t0 = time.time(); o = fn(); dt = time.time() - t0
    print('%-13s %.1f ms  %.0f spectra/s' % (name, dt * 1e3, S / dt))
a = run_single(); b = run_overlap()
vel = torch.empty_like(a[0]); res = torch.empty_like(a[1])
for k in range(NP):
    ix = torch.arange(k, S, NP, device=dev)
    vel[ix] = b[k][0]; res[ix] = b[k][1]
print('ccf vel equal', torch.equal(vel, a[0]), 'grid moments equal', torch.equal(res, a[1]))
