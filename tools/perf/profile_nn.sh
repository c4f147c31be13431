#!/bin/bash
# tools/perf/profile_nn.sh <tag>: BASELINE configs[3] (NN evaluator): bench line,
# kernel stats and the MFMA counters of nn_linear_kernel (separate --pmc run,
# kernel-trace only, the program directly after `--`).
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --evaluator nn > $R/gpurun_out/bench_nn_$tag.json 2> $R/gpurun_out/bench_nn_$tag.err
rm -rf /tmp/pnn_$tag /tmp/pmcnn_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pnn_$tag -o p -- python3 $R/bench.py --evaluator nn --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/pnn_$tag -name '*kernel_stats.csv' | head -1) $R/gpurun_out/kernel_stats_nn_$tag.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/pmcnn_$tag -o p -- python3 $R/bench.py --evaluator nn --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob, json, collections
f = glob.glob('/tmp/pmcnn_${tag}/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if 'nn_linear_kernel' not in r['Kernel_Name'] and 'nn_hidden_kernel' not in r['Kernel_Name']:
        continue
    k = r['Kernel_Name'].split('(')[0]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, v in agg.items():
    busy = sum(v['SQ_VALU_MFMA_BUSY_CYCLES']); gui = sum(v['GRBM_GUI_ACTIVE'])
    # per launch, median over the launches (one launch of a counter pass now and
    # then shows a GRBM_GUI_ACTIVE several times the others': first touch of pages)
    per = sorted(b / (g / 8 * 1024) for b, g in
                 zip(v['SQ_VALU_MFMA_BUSY_CYCLES'], v['GRBM_GUI_ACTIVE']) if g)
    out[k] = dict(launches=len(v['GRBM_GUI_ACTIVE']),
                  mfma_util_median_launch=per[len(per) // 2] if per else None,
                  mfma_util_per_launch=[round(x, 4) for x in per],
                  mfma_busy_cycles=busy, mfma_mops_f32=sum(v['SQ_INSTS_VALU_MFMA_MOPS_F32']),
                  grbm_gui_active_sum_over_8_xcds=gui,
                  # MfmaUtil of rocprofv3's derived metrics: busy cycles over
                  # (active cycles per XCD x 1024 SIMDs)
                  mfma_util=busy / (gui / 8 * 1024) if gui else None,
                  wave_cycles=sum(v['SQ_WAVE_CYCLES']), wait_inst_any=sum(v['SQ_WAIT_INST_ANY']),
                  wait_any=sum(v['SQ_WAIT_ANY']))
json.dump(out, open('gpurun_out/pmc_nn_${tag}.json', 'w'), indent=1)
print(json.dumps(out, indent=0)[:1200])
PY
tail -c 400 gpurun_out/bench_nn_$tag.json
