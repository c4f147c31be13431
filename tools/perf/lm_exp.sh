#!/bin/bash
# preprocess kernel time as a function of the LM iteration cap
cd $GRAFT_REPO_ROOT
for n in 1 4 8 60; do
  hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DRVS_LM_MAXIT=$n -c rvspecfit_amd/csrc/ccf.hip -o rvspecfit_amd/csrc/_build/ccf.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o rvspecfit_amd/librvsgpu.so rvspecfit_amd/csrc/_build/*.o
  python bench.py --spectra 2000 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print($n, d['kernels']['ccf_preprocess'], d['value'])"
done
