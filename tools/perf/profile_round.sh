#!/bin/bash
# tools/perf/profile_round.sh <tag>: default bench line, rocprofv3 kernel stats of
# the same command, and FETCH_SIZE / WRITE_SIZE counter passes (separate runs,
# kernel-trace only, the program directly after `--`).
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/bench_$tag.json 2> $R/gpurun_out/bench_$tag.err
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_$tag.log 2>&1
cp $(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1) $R/gpurun_out/kernel_stats_$tag.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_${tag}_$c.log 2>&1
done
cd $R
python3 - <<PY
import csv, glob, json, collections
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('/tmp/pmc_%s/**/*counter_collection.csv' % c, recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    out[c] = {k: (len(v), sum(v)) for k, v in agg.items()
              if any(x in k for x in ('chisq', 'ccf', 'spline', 'polylin', 'vsini', 'continuum', 'nn_'))}
json.dump(out, open('gpurun_out/pmc_${tag}_raw.json', 'w'), indent=1)

def per_launch(names, count_name):
    """(FETCH_SIZE x 2 + WRITE_SIZE) KB -> bytes, summed over the kernels of one
    library call, per launch of count_name (MI355X_MICROARCH.md, HBM section:
    gfx950 FETCH_SIZE reports half of wide coalesced reads, WRITE_SIZE exact)"""
    fe = sum(v[1] for k, v in out['FETCH_SIZE'].items() if any(n in k for n in names))
    wr = sum(v[1] for k, v in out['WRITE_SIZE'].items() if any(n in k for n in names))
    n = sum(v[0] for k, v in out['FETCH_SIZE'].items() if count_name in k)
    return dict(launches=n, fetch_size_raw_kb=fe, write_size_raw_kb=wr,
                hbm_bytes_per_launch=(2 * fe + wr) * 1024 / max(n, 1))
src = 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs of python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline, tag ${tag}'
t = {'ccf_xcorr': dict(per_launch(['ccf_xcorr_kernel'], 'ccf_xcorr_kernel'), source=src,
                       kernels='ccf_xcorr_kernel (one launch = one accumulator chunk x T templates x one arm)'),
     'chisq_grid': dict(per_launch(['chisq_grid_kernel'], 'chisq_grid_kernel<10, false>'), source=src,
                        kernels='chisq_grid_kernel<10,false> + <10,true> (one rvs_chisq_grid call = one arm of the batch)',
                        note='FETCH_SIZE doubling is calibrated for 16-B-per-lane streaming reads; this kernel gathers 32-B records and reads through the scalar cache')}
json.dump(t, open('gpurun_out/pmc_${tag}_traffic.json', 'w'), indent=1)
print(json.dumps(t, indent=0)[:1500])
PY
tail -c 600 gpurun_out/bench_$tag.json
