#!/bin/bash
# tools/perf/desi_fpb.sh: the DESI driver's 16-file line at files_per_batch = 4, 8, 16
# (alternating, in one job)
cd $GRAFT_REPO_ROOT
line() {
  timeout 300 python bench.py --spectra 2000 --steps 1 --warmup 1 --no-cpu-baseline --desi-file 500 --desi-nfiles ${NFILES:-16} "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); q=d['desi_file']
print(q['files'], q['files_per_batch'], q['fibres_per_s'], q['seconds'], q.get('stage_s'), q.get('fit_stage_s'))"
}
for rep in 1 2; do
  for fpb in ${FPB:-4 8 16}; do
    line --desi-files-per-batch $fpb
  done
done
