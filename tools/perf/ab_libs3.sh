#!/bin/bash
# tools/perf/ab_libs3.sh <name> ...: the optimiser lines with tools/perf/_bin/librvsgpu_<name>.so
# in place of the in-tree library ("head" = the in-tree one), alternating, in one job
cd $GRAFT_REPO_ROOT
cp rvspecfit_amd/librvsgpu.so /tmp/lib_head.so
for n in "$@"; do [ "$n" = head ] || cp tools/perf/_bin/librvsgpu_$n.so /tmp/lib_$n.so; done
line() {
  timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['process']
print(p['spectra'], p['spectra_per_s'], p['seconds'], p['stage_s'].get('bfgs'))"
}
for rep in 1 2; do
  for which in "$@"; do
    cp /tmp/lib_$which.so rvspecfit_amd/librvsgpu.so
    echo "== $which (rep $rep)"
    for n in ${SIZES:-500 2000}; do line --spectra $n --process $n; done
  done
done
cp /tmp/lib_head.so rvspecfit_amd/librvsgpu.so
