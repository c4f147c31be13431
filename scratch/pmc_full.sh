#!/bin/bash
# PMC passes on the DEFAULT bench command (10 000 spectra, T=76), one counter group per pass
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcfull_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmcfull_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json, collections
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('gpurun_out/pmcfull_%s/*/*counter_collection.csv' % c)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    out[c] = {k: (len(v), sum(v)) for k, v in agg.items() if any(x in k for x in ('chisq', 'ccf', 'spline', 'polylin', 'vsini'))}
json.dump(out, open('gpurun_out/pmcfull_summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
