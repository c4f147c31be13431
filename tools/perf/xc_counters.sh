#!/bin/bash
# SQ counters of ccf_xcorr_kernel (and the other big kernels) on the default bench
# step: which unit is busy.  Separate --pmc passes (a few counters each).
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-x}
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/xcc_$i -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --spectra 4000 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
nl = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('/tmp/xcc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if not any(s in k for s in ('ccf_xcorr', 'chisq_grid_kernel', 'ccf_preprocess', 'ccf_rfft')):
            continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        nl[k][r['Counter_Name']] += 1
out = {k: {c: v / nl[k][c] for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open('$R/gpurun_out/xc_counters_$tag.json', 'w'), indent=1)
for k, d in out.items():
    print(k)
    for c in sorted(d):
        print('   %-26s %.4g' % (c, d[c]))
PY
