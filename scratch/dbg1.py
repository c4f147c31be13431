import os, sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import GOLD, GOLD_CONFIG, gold_specdata
from rvspecfit_amd import spec_inter, spec_fit
from rvspecfit_amd.library import TemplateLibrary
cases = dict(np.load(os.path.join(GOLD, 'cases.npz')))
cfg = dict(GOLD_CONFIG); cfg['template_lib'] = 'golden://'
for n in ('gold_b', 'gold_r'):
    spec_inter.register_library(TemplateLibrary(n, np.load(os.path.join(GOLD, 'lib_%s.npz' % n))), 'golden://')
for tag in ['c0', 'c2']:
    sds = gold_specdata(cases, tag, spec_fit.SpecData)
    for i in (5,):
        k = '%s/chisq/t%d/' % (tag, i)
        val = spec_fit.get_chisq(sds, float(cases[k + 'vel']), tuple(cases[k + 'param']), None, options=dict(npoly=10), config=cfg)
        print(tag, i, repr(val), repr(float(cases[k + 'value'])), (val - cases[k + 'value']) / cases[k + 'value'])
tag = 'c1'
sds = gold_specdata(cases, tag, spec_fit.SpecData)
vg = cases['vel_grid']
for g in ('g1', 'g3'):
    k = '%s/%s/' % (tag, g)
    vs = float(cases[k + 'vsini']); rot = None if np.isnan(vs) else (vs,)
    pl = [tuple(_) for _ in cases[k + 'params_list']]
    b, _ = spec_fit.as_batch(sds)
    par = torch.as_tensor(np.array(pl))[None].to('cuda')
    vst = None if rot is None else torch.as_tensor([rot[0]], dtype=torch.float64).to('cuda')
    chisq, st, _ = spec_fit.chisq_grid_jobs(b, torch.as_tensor(vg).to('cuda'), par, vst, dict(npoly=10), cfg)
    got = chisq[0].cpu().numpy().T
    ref = cases[k + 'chisq_grid']
    print(g, got.shape, ref.shape, np.isnan(got).sum(), np.isnan(ref).sum(), st.cpu().numpy())
    bad = ~np.isfinite(got)
    print(np.nonzero(bad)[0][:10], np.nonzero(bad)[1][:10])
    ok = np.isfinite(got)
    print('max rel', np.max(np.abs(got[ok] - ref[ok]) / np.abs(ref[ok])))
