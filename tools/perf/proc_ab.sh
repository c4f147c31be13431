#!/bin/bash
# tools/perf/proc_ab.sh "<env assignments A>" "<env assignments B>": the optimiser lines
# under two settings of the package's environment switches, alternating, in one job
cd $GRAFT_REPO_ROOT
line() {
  timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['process']
print(p['spectra'], p['spectra_per_s'], p['seconds'])"
}
for rep in 1 2; do
  for which in "$1" "$2"; do
    echo "== [$which] (rep $rep)"
    for n in ${SIZES:-500 2000 10000}; do env $which bash -c "$(declare -f line); line --spectra $n --process $n"; done
  done
done
