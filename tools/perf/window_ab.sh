#!/bin/bash
# tools/perf/window_ab.sh: option nm_tail_window (rounds between two looks of the host in
# the optimiser's last rounds) 4 / 16 / 32 / 64, alternating, in one job
cd $GRAFT_REPO_ROOT
line() {
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python tools/perf/pp_line2.py | cut -c1-230
}
for rep in 1 2; do
  for m in 4 16 32 64; do
    echo "== nm_tail_window $m (rep $rep)"
    export RVS_NM_TAIL_WINDOW=$m
    line --spectra 10000 --process 10000
    line --spectra 2000 --process 2000
    line --spectra 2000 --process 500
  done
done
