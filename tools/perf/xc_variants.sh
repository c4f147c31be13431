#!/bin/bash
# builds librvsgpu with variants of ccf_fft.hip in turn (XC_VARIANTS: hipcc -D
# flags, one variant per word) and prints the default step's xcorr time
cd $GRAFT_REPO_ROOT
cp rvspecfit_amd/librvsgpu.so /tmp/librvsgpu_orig.so
trap 'cp /tmp/librvsgpu_orig.so rvspecfit_amd/librvsgpu.so' EXIT
for v in ${XC_VARIANTS:--DXC_GROUP_BYTES=1048576 -DXC_GROUP_BYTES=2097152}; do
  hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 $v -c rvspecfit_amd/csrc/ccf_fft.hip -o /tmp/xc_v.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o rvspecfit_amd/librvsgpu.so /tmp/xc_v.o $(ls rvspecfit_amd/csrc/_build/*.o | grep -v /ccf_fft.o)
  python -m pytest tests -x -q -m gpu -k "ccf or xcorr" 2>&1 | tail -1
  python bench.py ${XC_BENCH_ARGS:---steps 5} --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['kernels']['ccf_xcorr'])"
done
