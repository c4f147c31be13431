#!/bin/bash
# real HBM bytes of the polylinear gather on a small (cache resident) and a large
# library: FETCH_SIZE / WRITE_SIZE passes of tools/perf/gather_bench.py
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for grid in 7,7,7,7 40,11,8,5; do
  python3 $R/tools/perf/gather_bench.py $grid 10000 b
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/gp_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/gp_$c -o p -- \
      python3 $R/tools/perf/gather_bench.py $grid 10000 b > /tmp/gp_$c.log 2>&1
  done
  python3 - <<PY
import csv, glob
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('/tmp/gp_%s/**/*counter_collection.csv' % c, recursive=True)[0]
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(f))
         if 'polylinear_kernel' in r['Kernel_Name']]
    out[c] = (len(v), sum(v) / max(len(v), 1) * 1024)
print('  grid $grid: polylinear_kernel launches %d; per launch FETCH_SIZE raw %.3f GB '
      '(x2 per the gfx950 note: %.3f GB), WRITE_SIZE %.3f GB'
      % (out['FETCH_SIZE'][0], out['FETCH_SIZE'][1] / 1e9,
         2 * out['FETCH_SIZE'][1] / 1e9, out['WRITE_SIZE'][1] / 1e9))
PY
done
