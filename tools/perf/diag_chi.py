"""diagnostic: where does the full-size chi^2 difference device vs oracle come from"""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import numpy as np, torch
import bench
from oracle import rvs_oracle as orc
from rvspecfit_amd import engine, spec_inter, spec_fit, pipeline
from rvspecfit_amd.library import TemplateLibrary
from chisq_truth import chisq0_longdouble
dev = torch.device('cuda', 0)
def gpu_convolve(lam, templ, vsini):
    t = torch.as_tensor(np.ascontiguousarray(templ)).to(dev)
    v = torch.as_tensor(np.ascontiguousarray(vsini)).to(dev)
    return engine.convolve_vsini(lam, t, v).cpu().numpy()
dicts = bench.build_library_dicts(64, gpu_convolve)
for name, d in dicts.items():
    spec_inter.register_library(TemplateLibrary(name, d, device=dev), bench.CONFIG['template_lib'])
olibs = {k: orc.make_library(v) for k, v in dicts.items()}
S = 64
tp = bench.truth_params(S, seed=3)
arms = bench.make_spectra_device(tp, dev)
batch = engine.SpecBatch([engine.ArmData(n, lam, sp, es, bad, device=dev) for n, lam, sp, es, bad in arms])
rec = pipeline.fit_batch(batch, bench.CONFIG, options=bench.OPTIONS).cpu().numpy()
F = pipeline.RECORD_FIELDS
vg = np.arange(-1000, 1000, 5.)
worst = []
for i in range(S):
    sds = [orc.SpecData(n, lam, sp[i].cpu().numpy(), es[i].cpu().numpy(), badmask=bad[i].cpu().numpy() != 0) for n, lam, sp, es, bad in arms]
    o = orc.ccf_fit(sds, bench.CONFIG, olibs)
    vs = o['best_vsini']; rot = None if np.isnan(vs) else (vs,)
    grid = orc.chisq_grid_fast(sds, vg, o['best_par'], rot, bench.OPTIONS, bench.CONFIG, olibs)
    s = orc.grid_summary(vg, grid[:, None])
    d = rec[i, F.index('best_chi')] - s['best_chi']
    worst.append((abs(d), i, d, s['best_chi'], vs, tp['snr'][i], s['best_vel']))
worst.sort(reverse=True)
for w in worst[:6]:
    print('dchi %.3e spec %d  d=%.3e chi=%.3f vsini=%s snr=%.1f vel=%.2f' % w)
# per-arm at the worst spectrum, at its best grid velocity
_, i, _, _, vs, _, bv = worst[0]
sds = [orc.SpecData(n, lam, sp[i].cpu().numpy(), es[i].cpu().numpy(), badmask=bad[i].cpu().numpy() != 0) for n, lam, sp, es, bad in arms]
gsds = [spec_fit.SpecData(n, lam, sp[i].cpu().numpy(), es[i].cpu().numpy(), badmask=bad[i].cpu().numpy() != 0) for n, lam, sp, es, bad in arms]
o = orc.ccf_fit(sds, bench.CONFIG, olibs)
rot = None if np.isnan(o['best_vsini']) else (o['best_vsini'],)
vel = float(vg[np.argmin(np.abs(vg - bv))])
for k in range(3):
    oc = orc.get_chisq([sds[k]], vel, tuple(o['best_par']), rot, options=bench.OPTIONS, config=bench.CONFIG, libs=olibs)
    gc = spec_fit.get_chisq([gsds[k]], vel, tuple(o['best_par']), rot, options=bench.OPTIONS, config=bench.CONFIG)
    lib = olibs[sds[k].name]
    outside, tspec = orc.get_cur_templ(lib, tuple(o['best_par']), rot)
    spl = orc.Spline(lib.lam, tspec, log_step=lib.log_step)
    ev = orc.eval_rv(spl, vel, sds[k].lam)
    polys = orc.get_poly_basis(sds[k].lam, 10)
    ld = float(chisq0_longdouble(sds[k].spec, ev, polys, sds[k].espec))
    # device template
    gt = spec_fit.getCurTempl(sds[k].name, tuple(o['best_par']), rot, bench.CONFIG)
    dt = np.max(np.abs(np.asarray(gt[2]) - tspec) / np.abs(tspec))
    print(sds[k].name, 'oracle %.6f device %.6f longdouble(oracle templ) %.6f  d(dev-orc) %.3e d(orc-ld) %.3e  templ rel diff %.2e' % (oc, gc, ld, gc - oc, oc - ld, dt))
