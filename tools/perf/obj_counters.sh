#!/bin/bash
# SQ / TCC counters of the fused objective kernel on tools/perf/obj_bench (built
# here with OBJ_FLAGS): which unit is busy, and the fabric-side bytes per launch.
# Separate --pmc passes (a few counters each; no trace domains besides the kernel
# trace).  tools/perf/obj_counters.sh <tag> [bench args ...]
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-x}; shift
args="${@:-9000 2}"
mkdir -p $R/tools/perf/_bin $R/gpurun_out
hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-unused-value -Wno-unused-result -DOBJ_ONLY_P=${OBJ_P:-10} $OBJ_FLAGS \
  -o $R/tools/perf/_bin/obj_bench_c $R/tools/perf/obj_bench.hip \
  -L$R/rvspecfit_amd -l:librvsgpu.so -Wl,-rpath,$R/rvspecfit_amd 2>/dev/null || { echo build failed; exit 1; }
rm -rf /tmp/objc_*
i=0
if [ -n "$OBJ_ONLY_TRAFFIC" ]; then SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"); else SETS=(
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"); fi
for set in "${SETS[@]}"; do
  i=$((i+1))
  rm -rf /tmp/objc_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/objc_$i -o p -- $R/tools/perf/_bin/obj_bench_c $args > /tmp/objc_$i.log 2>&1
done
rm -rf /tmp/objc_t
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/objc_t -o p -- $R/tools/perf/_bin/obj_bench_c $args > /tmp/objc_t.log 2>&1
python3 - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
nl = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('/tmp/objc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'objective' not in k:
            continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        nl[k][r['Counter_Name']] += 1
out = {k: {c: v / nl[k][c] for c, v in d.items()} for k, d in acc.items()}
dur = {}
for f in glob.glob('/tmp/objc_t/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Name'].split('(')[0]] = (float(r['AverageNs']), int(r['Calls']))
for k, d in out.items():
    t = dur.get(k, (None, 0))
    d['avg_duration_ns'], d['calls'] = t
    if d.get('GRBM_GUI_ACTIVE'):
        cyc = d['GRBM_GUI_ACTIVE'] / 8          # summed over the 8 XCDs
        d['valu_busy'] = round(d.get('SQ_ACTIVE_INST_VALU', 0) * 4 / (cyc * 1024), 4)
        d['lds_busy'] = round(d.get('SQ_ACTIVE_INST_LDS', 0) * 4 / (cyc * 1024), 4)
        d['vmem_busy'] = round(d.get('SQ_ACTIVE_INST_VMEM', 0) * 4 / (cyc * 1024), 4)
    if d.get('TCC_HIT_sum') is not None and d.get('TCC_MISS_sum') is not None:
        d['l2_hit_rate'] = round(d['TCC_HIT_sum'] / max(1.0, d['TCC_HIT_sum'] + d['TCC_MISS_sum']), 4)
line = [l for l in open('/tmp/objc_t.log') if l.startswith('J ')]
json.dump(dict(args='$args', flags='$OBJ_FLAGS', bench_line=line[-1].strip() if line else None,
               kernels=out), open('$R/gpurun_out/obj_counters_$tag.json', 'w'), indent=1)
for k, d in out.items():
    print(k)
    for c in sorted(d):
        print('   %-26s %s' % (c, d[c]))
PY
