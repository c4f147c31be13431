#!/bin/bash
# SQ counters of the NN kernels on tools/perf/nn_bench (B rows, default 10000), one
# rocprofv3 pass per group of counters: tools/perf/nn_sq.sh <tag> [B]
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-x}; B=${2:-10000}
mkdir -p $R/tools/perf/_bin $R/gpurun_out
hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-unused-value -Wno-unused-result $NN_FLAGS -I $R/include \
  -o $R/tools/perf/_bin/nn_bench_c $R/tools/perf/nn_bench.hip 2>/dev/null || { echo build failed; exit 1; }
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  rm -rf /tmp/nnsq_$i
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/nnsq_$i -o p -- $R/tools/perf/_bin/nn_bench_c $B > /tmp/nnsq_$i.log 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/nnsq_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'nn_' not in r['Kernel_Name']:
            continue
        agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: sum(v) / len(v) for c, v in sorted(d.items())} for k, d in agg.items()}
json.dump(dict(rows=$B, flags='$NN_FLAGS', per_launch=out), open('$R/gpurun_out/nn_sq_$tag.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
