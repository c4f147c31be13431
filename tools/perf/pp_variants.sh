#!/bin/bash
# builds librvsgpu with LM_PIX = 1, 3, 6 in turn and prints the preprocess time
cd $GRAFT_REPO_ROOT
cp rvspecfit_amd/librvsgpu.so /tmp/librvsgpu_orig.so
trap 'cp /tmp/librvsgpu_orig.so rvspecfit_amd/librvsgpu.so' EXIT
for lp in 1 3 6; do
  hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DLM_PIX=$lp -c rvspecfit_amd/csrc/ccf.hip -o /tmp/ccf_v.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o rvspecfit_amd/librvsgpu.so /tmp/ccf_v.o $(ls rvspecfit_amd/csrc/_build/*.o | grep -v /ccf.o)
  python bench.py --steps 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LM_PIX $lp', d['value'], d['kernels']['ccf_preprocess'])"
done
