"""Reduce one configuration's FETCH_SIZE / WRITE_SIZE passes to HBM bytes per launch
of ccf_xcorr_kernel and of one rvs_chisq_grid call, and merge them into a JSON file
keyed by the configuration (bench.py's config.traffic_key, read from the bench line
the profiled run printed).  Corrections as MI355X_MICROARCH.md's HBM section
prescribes: counters in KB; on gfx950 FETCH_SIZE reports half of the bytes of wide
coalesced reads -> doubled; WRITE_SIZE exact.

usage: pmc_traffic.py OUT.json FETCH_DIR WRITE_DIR BENCH_LOG TAG "bench args"
"""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(d):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    return agg


def main():
    out, fdir, wdir, log, tag, bargs = sys.argv[1:7]
    key = None
    for ln in open(log):
        if ln.lstrip().startswith('{'):
            key = json.loads(ln)['config']['traffic_key']
    if key is None:
        raise SystemExit('no bench line in ' + log)
    fe, wr = per_kernel(fdir), per_kernel(wdir)

    def entry(names, count_name, what):
        f = sum(sum(v) for k, v in fe.items() if any(n in k for n in names))
        w = sum(sum(v) for k, v in wr.items() if any(n in k for n in names))
        n = sum(len(v) for k, v in fe.items() if all(c in k for c in count_name))
        return dict(launches=n, fetch_size_raw_kb=f, write_size_raw_kb=w,
                    hbm_bytes_per_launch=(2 * f + w) * 1024 / max(n, 1),
                    fetch_uncorrected_bytes_per_launch=f * 1024 / max(n, 1),
                    kernels=what,
                    source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, '
                           'separate runs of python3 bench.py --steps 1 --warmup '
                           '1 --no-cpu-baseline %s, tag %s' % (bargs, tag))
    d = json.load(open(out)) if os.path.exists(out) else {}
    d[key] = dict(
        ccf_xcorr=entry(['ccf_xcorr_'], ('ccf_xcorr_', ),
                        'ccf_xcorr_ws_kernel / ccf_xcorr_kernel (one launch = one '
                        'accumulator chunk x T templates x one arm)'),
        chisq_grid=entry(['chisq_grid_kernel', 'chisq_grid_resol'],
                         # (one full-wave launch per rvs_chisq_grid call, any npoly)
                         ('chisq_grid_kernel<', ', false>')
                         if not any('chisq_grid_resol' in k for k in fe)
                         else ('chisq_grid_resol', ),
                         'every chisq_grid kernel of one rvs_chisq_grid call (full '
                         'waves + packed left-over velocities); FETCH_SIZE '
                         'doubling is calibrated for 16-B streaming reads, this '
                         'kernel gathers 32-B records'))
    json.dump(d, open(out, 'w'), indent=1)
    print(key, {k: round(v['hbm_bytes_per_launch'] / 1e9, 3) for k, v in d[key].items()})


if __name__ == '__main__':
    main()
