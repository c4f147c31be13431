#!/bin/bash
# tools/perf/spec_ab.sh: the optimiser's last rounds as one launch per round
# (option nm_spec_max) against two, alternating, in one job
cd $GRAFT_REPO_ROOT
line() {
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python tools/perf/pp_line2.py | cut -c1-260
}
for rep in 1 2; do
  for m in 0 21 40; do
    echo "== nm_spec_max $m (rep $rep)"
    export RVS_NM_SPEC_MAX=$m
    line --spectra 10000 --process 10000
    line --spectra 2000 --process 2000
    line --spectra 2000 --process 500
    line --spectra 2000 --desi-file 500
  done
done
