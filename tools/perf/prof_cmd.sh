#!/bin/bash
# tools/perf/prof_cmd.sh <tag> <bench args...>: rocprofv3 kernel stats of a bench.py invocation
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python3 $R/bench.py "$@" > $R/gpurun_out/prof_$tag.log 2>&1
cp $(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1) $R/gpurun_out/kernel_stats_$tag.csv
python3 - <<PY
import csv
for i, r in enumerate(csv.DictReader(open("$R/gpurun_out/kernel_stats_$tag.csv"))):
    if i < 14:
        print(r["Name"][:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
tail -1 $R/gpurun_out/prof_$tag.log | cut -c1-300
