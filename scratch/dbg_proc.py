import os, sys, numpy as np, torch, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import GOLD, GOLD_CONFIG
from rvspecfit_amd import spec_inter, spec_fit, vel_fit, engine
from rvspecfit_amd.library import TemplateLibrary
cfg = dict(GOLD_CONFIG); cfg['template_lib'] = 'golden://'
for n in ('gold_b', 'gold_r'):
    spec_inter.register_library(TemplateLibrary(n, np.load(os.path.join(GOLD, 'lib_%s.npz' % n))), 'golden://')
cases = dict(np.load(GOLD + '/cases.npz'))
g = dict(np.load(GOLD + '/process_cases.npz'))
def sds(tag):
    names = [str(_) for _ in cases[tag + '/names']]
    return [spec_fit.SpecData(n, cases['%s/%s/lam' % (tag, n)], cases['%s/%s/spec' % (tag, n)], cases['%s/%s/espec' % (tag, n)], badmask=cases['%s/%s/badmask' % (tag, n)]) for n in names]
t = sys.argv[1] if len(sys.argv) > 1 else 'p1'
pd0 = dict(zip([str(_) for _ in g[t + '/start_keys']], [float(_) for _ in g[t + '/start_vals']]))
fix = [str(_) for _ in g[t + '/fix']]
sd = sds(str(g[t + '/case']))
batch, _ = spec_fit.as_batch(sd)
idx = torch.zeros(1, dtype=torch.long, device='cuda')
x = torch.as_tensor(g[t + '/nm_x'])[None].to('cuda')
names = ['teff', 'logg', 'feh', 'alpha']
par = torch.as_tensor(g[t + '/param'])[None].to('cuda')
vs = None if not np.isfinite(g[t + '/vsini']) else torch.as_tensor([float(g[t + '/vsini'])]).to('cuda')
vel = torch.as_tensor([float(g[t + '/nm_x'][0])]).to('cuda')
c, st = spec_fit.chisq_jobs(batch, idx, vel, par, vs, dict(npoly=10), cfg, vel_bounds=(-1000, 1000))
print('chisq_jobs', c, st, 'golden nm_fun', g[t + '/nm_fun'])
from rvspecfit_amd import neldermead
_orig = neldermead.minimize
def dbg_min(func, simplex, **kw):
    cnt = [0]
    def f2(idx, X):
        v = func(idx, X)
        cnt[0] += 1
        if cnt[0] < 3 or not torch.isfinite(v).all():
            if cnt[0] < 40:
                print('eval', cnt[0], X.cpu().numpy(), v.cpu().numpy())
        return v
    return _orig(f2, simplex, **kw)
neldermead.minimize = dbg_min
tm = {}
t0 = time.time()
r = vel_fit.process(sd, pd0, fixParam=fix, options=dict(npoly=10), config=cfg, timers=tm)
print('time', time.time() - t0, tm)
for k in ('vel', 'vel_err', 'chisq', 'param', 'param_err', 'nm_nit', 'nm_nfev', 'nm_rounds', 'minimize_success', 'bad_hessian', 'vsini'):
    print(k, r.get(k), g.get(t + '/' + k))
