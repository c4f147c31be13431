#!/bin/bash
# tools/perf/desi_ab.sh "<env A>" "<env B>": the DESI driver's 16-file line under two
# settings of the package's environment switches, alternating, in one job
cd $GRAFT_REPO_ROOT
line() {
  timeout 300 python bench.py --spectra 2000 --steps 1 --warmup 1 --no-cpu-baseline --desi-file 500 --desi-nfiles ${NFILES:-16} "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); q=d['desi_file']
print(q['files'], q['fibres_per_s'], q['seconds'], q['stage_s'])"
}
for rep in 1 2; do
  for which in "$1" "$2"; do
    echo "== [$which] (rep $rep)"
    env $which bash -c "$(declare -f line); line"
  done
done
