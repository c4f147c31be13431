#!/bin/bash
# tools/perf/ab_libs.sh: the in-tree librvsgpu.so against tools/perf/_bin/librvsgpu_base.so
# (another build of the same ABI) on the optimiser lines, alternating, in one job.
cd $GRAFT_REPO_ROOT
cp rvspecfit_amd/librvsgpu.so /tmp/lib_new.so
cp tools/perf/_bin/librvsgpu_base.so /tmp/lib_base.so
line() {
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python tools/perf/pp_line2.py
}
for rep in 1 2; do
  for which in base new; do
    cp /tmp/lib_$which.so rvspecfit_amd/librvsgpu.so
    echo "== $which (rep $rep)"
    line --spectra 10000 --process 10000
    line --spectra 2000 --process 2000
    line --spectra 2000 --process 500
    line --spectra 2000 --desi-file 500
  done
done
cp /tmp/lib_new.so rvspecfit_amd/librvsgpu.so
