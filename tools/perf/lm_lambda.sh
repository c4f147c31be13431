#!/bin/bash
# preprocess kernel time and CCF parity as a function of the initial LM damping
cd $GRAFT_REPO_ROOT
cp rvspecfit_amd/librvsgpu.so /tmp/orig.so
for n in 1e-3 1e-4 1e-5 1e-7; do
  hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DRVS_LM_LAMBDA0=$n -c rvspecfit_amd/csrc/ccf.hip -o /tmp/ccf_l.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o rvspecfit_amd/librvsgpu.so /tmp/ccf_l.o $(ls rvspecfit_amd/csrc/_build/*.o | grep -v /ccf.o)
  python bench.py --spectra 4000 --steps 2 --warmup 1 --cpu-sample 32 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['parity_sample']; print('$n', d['kernels']['ccf_preprocess'], d['value'], p['best_id_equal'], p['max_abs_dvrad_ccf'])"
done
cp /tmp/orig.so rvspecfit_amd/librvsgpu.so
