#!/bin/bash
# MFMA counters of the wide last layer (nn_linear_kernel<true,...>) on
# tools/perf/nn_bench at B rows (default 10000): tools/perf/nn_counters.sh <tag> [B]
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-x}; B=${2:-10000}
mkdir -p $R/tools/perf/_bin $R/gpurun_out
hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-unused-value -Wno-unused-result $NN_FLAGS -I $R/include \
  -o $R/tools/perf/_bin/nn_bench_c $R/tools/perf/nn_bench.hip 2>/dev/null || { echo build failed; exit 1; }
rm -rf /tmp/nnc_1 /tmp/nnc_t
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/nnc_1 -o p -- $R/tools/perf/_bin/nn_bench_c $B > /tmp/nnc_1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nnc_t -o p -- $R/tools/perf/_bin/nn_bench_c $B > /tmp/nnc_t.log 2>&1
python3 - <<PY
import csv, glob, json, collections
f = glob.glob('/tmp/nnc_1/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if 'nn_' not in r['Kernel_Name']:
        continue
    agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
dur = {}
for f in glob.glob('/tmp/nnc_t/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Name'].split('(')[0]] = (float(r['AverageNs']), int(r['Calls']))
out = {}
for k, v in agg.items():
    per = sorted(b / (g / 8 * 1024) for b, g in
                 zip(v['SQ_VALU_MFMA_BUSY_CYCLES'], v['GRBM_GUI_ACTIVE']) if g)
    out[k] = dict(launches=len(per), mfma_util_median_launch=per[len(per) // 2] if per else None,
                  mfma_util=sum(v['SQ_VALU_MFMA_BUSY_CYCLES']) / (sum(v['GRBM_GUI_ACTIVE']) / 8 * 1024),
                  mfma_mops_f32=sum(v['SQ_INSTS_VALU_MFMA_MOPS_F32']) / max(1, len(per)),
                  wave_cycles=sum(v['SQ_WAVE_CYCLES']) / max(1, len(per)),
                  wait_inst_any=sum(v['SQ_WAIT_INST_ANY']) / max(1, len(per)),
                  avg_duration_ns=dur.get(k, (None, 0))[0])
line = [l for l in open('/tmp/nnc_t.log') if 'TFLOP/s' in l]
json.dump(dict(rows=$B, flags='$NN_FLAGS', bench_line=line[-1].strip()[-160:] if line else None, kernels=out),
          open('$R/gpurun_out/nn_counters_$tag.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
print(line[-1].strip()[-160:] if line else '')
PY
