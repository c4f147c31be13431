#!/bin/bash
# tools/perf/ab_step.sh: the in-tree librvsgpu.so against tools/perf/_bin/librvsgpu_base.so
# (another build of the same ABI) on the contract step, alternating, in one job:
# spectra/s, ms per step, the cross-correlation's and the chi^2 grid's ms per step
cd $GRAFT_REPO_ROOT
cp rvspecfit_amd/librvsgpu.so /tmp/lib_new.so
cp tools/perf/_bin/librvsgpu_base.so /tmp/lib_base.so
line() {
  timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print(round(d['value']), d['ms_per_step'], 'xcorr', k['ccf_xcorr']['ms_per_step'], 'grid', k['chisq_grid']['ms_per_step'], 'prep', k['ccf_preprocess']['ms_per_step'])"
}
for rep in 1 2; do
  for which in base new; do
    cp /tmp/lib_$which.so rvspecfit_amd/librvsgpu.so
    echo "== $which (rep $rep)"
    line "$@"
  done
done
cp /tmp/lib_new.so rvspecfit_amd/librvsgpu.so
