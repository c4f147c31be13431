import sys, os, time, json, subprocess, numpy as np
sys.path.insert(0,'.')
import bench
from rvspecfit_amd import synth
n=4
tp=bench.truth_params(n, 3)
rng=np.random.RandomState(5)
sample=dict(n=n)
for a in bench.ARMS:
    lam=bench.obs_lam(a)
    wres=0.5*sum(synth.DESI_ARMS[a]['templ'][:2])/bench.RESOL/2.35
    sp0=synth.spectra_batch(lam,tp['teff'],tp['logg'],tp['feh'],tp['alpha'],vel=tp['vel'],wresol=wres)
    es=sp0/tp['snr'][:,None]
    spec=sp0+es*rng.standard_normal(sp0.shape)
    bad=rng.uniform(size=sp0.shape)<0.05
    es=np.where(bad,es*1e4,es)
    sample['spec_'+a]=spec; sample['espec_'+a]=es; sample['bad_'+a]=bad.astype(np.uint8)
np.savez('/tmp/sample.npz',**sample)
t=time.time()
out=subprocess.run([sys.executable,'bench.py','--cpu-worker','/tmp/sample.npz','--cpu-cores','4'],stdout=subprocess.PIPE,stderr=subprocess.PIPE,text=True)
print(time.time()-t, out.stderr[-3000:])
r=json.loads(out.stdout.strip().splitlines()[-1])
print({k:v for k,v in r.items() if k!='recs'})
for i,x in enumerate(r['recs']): print(x[:5], tp['vel'][i], tp['snr'][i])
