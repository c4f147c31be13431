import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
from rvspecfit_amd import engine, spec_inter, spec_fit, pipeline
from rvspecfit_amd.library import TemplateLibrary
from oracle import rvs_oracle as orc
dev = torch.device('cuda')
def gpu_convolve(lam, templ, vsini):
    t = torch.as_tensor(np.ascontiguousarray(templ)).to(dev); v = torch.as_tensor(np.ascontiguousarray(vsini)).to(dev)
    return engine.convolve_vsini(lam, t, v).cpu().numpy()
dicts = bench.build_library_dicts(64, gpu_convolve)
for n, d in dicts.items():
    spec_inter.register_library(TemplateLibrary(n, d, device=dev), bench.CONFIG['template_lib'])
olibs = {k: orc.Library(v) for k, v in dicts.items()}
S = 64
tp = bench.truth_params(2000, 3)
tp = {k: (v[:S] if k != 'seed' else v) for k, v in tp.items()}
arms = bench.make_spectra_device(tp, dev)
batch = engine.SpecBatch([engine.ArmData(n, lam, sp, es, bad, device=dev) for n, lam, sp, es, bad in arms])
rec = pipeline.fit_batch(batch, bench.CONFIG, options=bench.OPTIONS).cpu().numpy()
vg = np.arange(-1000, 1000, 5.)
worst = (0, -1)
for i in range(S):
    sds = [orc.SpecData(n, lam, sp[i].cpu().numpy(), es[i].cpu().numpy(), badmask=bad[i].cpu().numpy() != 0) for n, lam, sp, es, bad in arms]
    o = orc.ccf_fit(sds, bench.CONFIG, olibs)
    vs = o['best_vsini']; rot = None if np.isnan(vs) else (vs,)
    grid = orc.chisq_grid_fast(sds, vg, o['best_par'], rot, bench.OPTIONS, bench.CONFIG, olibs)
    s = orc.grid_summary(vg, grid[:, None])
    d = abs(rec[i, 11] - s['best_chi'])
    if d > worst[0]: worst = (d, i, o, rot, s, sds)
d, i, o, rot, s, sds = worst
print('worst', i, 'dchi', d, 'chi', s['best_chi'], 'snr', tp['snr'][i], 'par', o['best_par'], 'vsini', rot, 'best_vel', s['best_vel'], rec[i, 7])
v = float(vg[np.argmin(np.abs(vg - s['best_vel']))])
for a, sd in zip(bench.ARMS, sds):
    osd = [sd]
    gsd = [spec_fit.SpecData(sd.name, sd.lam, sd.spec, sd.espec, badmask=sd.badmask)]
    oc = orc.get_chisq(osd, v, o['best_par'], rot, options=bench.OPTIONS, config=bench.CONFIG, libs=olibs, use_c=True)
    gc = spec_fit.get_chisq(gsd, v, tuple(o['best_par']), rot, options=bench.OPTIONS, config=bench.CONFIG)
    oo, ot = orc.get_cur_templ(olibs[sd.name], o['best_par'], rot)
    go, lam_t, gt, _, _ = spec_fit.getCurTempl(sd.name, tuple(o['best_par']), rot, bench.CONFIG)
    print(a, 'chisq orc', oc, 'gpu', gc, 'diff', gc - oc, 'outside', oo, go, 'templ max rel diff', np.abs(gt / ot - 1).max())
