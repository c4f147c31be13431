#!/bin/bash
# tools/perf/resol_counters.sh: SQ counters and kernel statistics of the velocity-grid kernel
# with resolution matrices (bench.py --resolution-matrix), separate --pmc passes
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1)); rm -rf /tmp/rc_$i
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/rc_$i -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --resolution-matrix > /tmp/rc_$i.log 2>&1
done
rm -rf /tmp/rc_t
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rc_t -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --resolution-matrix > /tmp/rc_t.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nl=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('/tmp/rc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'resol' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); nl[k][r['Counter_Name']] += 1
for k,v in acc.items():
    n=nl[k]['SQ_WAVES']; print(k, 'launches', n)
    for c,x in v.items(): print('   ', c, x/max(1,nl[k][c]))
    if 'SQ_ACTIVE_INST_VALU' in v: print('    valu_busy', v['SQ_ACTIVE_INST_VALU']/v['SQ_BUSY_CYCLES']*4/ (256*4) if False else '')
for f in glob.glob('/tmp/rc_t/**/*kernel_stats.csv', recursive=True):
    for i,r in enumerate(csv.DictReader(open(f))):
        if i<5: print(r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
PY
