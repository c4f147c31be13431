#!/bin/bash
# kernel-time summary of the optimiser stage: tools/perf/prof_proc.sh <nspectra>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rm -rf /tmp/prof_proc
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_proc -o p -- python3 tools/perf/proc_time.py $1 > gpurun_out/prof_proc.log 2>&1
f=$(find /tmp/prof_proc -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/prof_proc_kernel_stats.csv
python3 - <<PY
import csv
for i, r in enumerate(csv.DictReader(open("gpurun_out/prof_proc_kernel_stats.csv"))):
    if i < 12:
        print(r["Name"][:50], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
grep "^S " gpurun_out/prof_proc.log
