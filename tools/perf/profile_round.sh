#!/bin/bash
# tools/perf/profile_round.sh <tag>: default bench line, rocprofv3 kernel stats of
# the same command, and FETCH_SIZE / WRITE_SIZE counter passes (separate runs).
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/bench_$tag.json 2> $R/gpurun_out/bench_$tag.err
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_$tag.log 2>&1
cp $(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1) $R/gpurun_out/kernel_stats_$tag.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_${tag}_$c.log 2>&1
done
cd $R
python3 - <<PY
import csv, glob, json, collections
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('/tmp/pmc_%s/**/*counter_collection.csv' % c, recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    out[c] = {k: (len(v), sum(v)) for k, v in agg.items()
              if any(x in k for x in ('chisq', 'ccf', 'spline', 'polylin', 'vsini', 'continuum'))}
json.dump(out, open('gpurun_out/pmc_${tag}_raw.json', 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items()}, indent=0)[:1500])
PY
tail -c 600 gpurun_out/bench_$tag.json
