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
  rm -rf /tmp/xcc_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/xcc_$i -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline $XC_ARGS > /tmp/xcc_$i.log 2>&1
done
# kernel durations of the same command (for rates per second)
rm -rf /tmp/xcc_t
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/xcc_t -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline $XC_ARGS > /tmp/xcc_t.log 2>&1
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
# average kernel duration (ns) from the stats pass
dur = {}
for f in glob.glob('/tmp/xcc_t/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Name'].split('(')[0]] = float(r['AverageNs'])
key = None
for ln in open('/tmp/xcc_1.log'):
    if ln.lstrip().startswith('{'):
        key = json.loads(ln)['config']['traffic_key']
for k, d in out.items():
    t = dur.get(k)
    d['avg_duration_ns'] = t
    # 256 CUs x 4 SIMDs; SQ_ACTIVE_INST_VALU counts busy cycles summed over SIMDs
    # (per quad-cycle on this chip, calibrated in round 2: x4 / (GUI cycles x 1024))
    if d.get('GRBM_GUI_ACTIVE'):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        cyc = d['GRBM_GUI_ACTIVE'] / 8
        d['valu_busy'] = round(d.get('SQ_ACTIVE_INST_VALU', 0) * 4 / (cyc * 1024), 4)
        d['lds_busy'] = round(d.get('SQ_ACTIVE_INST_LDS', 0) * 4 / (cyc * 1024), 4)
    if t and d.get('SQ_WAVES') and 'xcorr' in k:
        # L2 -> L1 bytes of a block (8 waves): the four operand arrays
        # (nfft/2+1 complex128 each) + the LDS twiddle rows; nfft from the key
        nfft = int([x for x in key.split('|') if x.startswith('nfft=')][0][5:])
        per_block = 4 * (nfft // 2 + 1) * 16 + (nfft // 16) * 16
        d['l2_to_l1_TBps'] = round(d['SQ_WAVES'] / 8 * per_block / (t * 1e-9) / 1e12, 2)
        if 'xcorr_ws' in k:
            # the persistent form: a block walks T templates; every vector load is
            # 16 B per lane (operands, fold twiddles, the spectrum once per block)
            d['l2_to_l1_TBps'] = round(d.get('SQ_INSTS_VMEM_RD', 0) * 64 * 16 / (t * 1e-9) / 1e12, 2)
        # cross-check: vector-memory read instructions x 64 lanes x 16 B (upper
        # bound: the table / mask loads are narrower)
        d['vmem_rd_TBps_upper'] = round(d.get('SQ_INSTS_VMEM_RD', 0) * 64 * 16 / (t * 1e-9) / 1e12, 2)
json.dump(dict(traffic_key=key, args='$XC_ARGS', kernels=out),
          open('$R/gpurun_out/xc_counters_$tag.json', 'w'), indent=1)
for k, d in out.items():
    print(k)
    for c in sorted(d):
        print('   %-26s %s' % (c, d[c]))
PY
