#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in "" "-DRVS_XC_SKIPFFT"; do
  hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 $f -c rvspecfit_amd/csrc/ccf_fft.hip -o rvspecfit_amd/csrc/_build/ccf_fft.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o rvspecfit_amd/librvsgpu.so rvspecfit_amd/csrc/_build/*.o
  python bench.py --spectra 2000 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['kernels']['ccf_xcorr'], d['value'])"
done
