"""GPU occupancy of the optimiser stage from a rocprofv3 kernel trace of
`bench.py --process N`: for every vel_fit.process call (runs of kernels without a gap
of more than 30 ms) the span, the time some kernel runs, the time an objective kernel
runs, the time two run side by side, the sum of the objective kernels' durations.
usage: proc_busy.py p_kernel_trace.csv"""
import csv
import sys


def union(iv):
    tot, cs, ce = 0, None, None
    for s, e in sorted(iv):
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)


def overlap2(iv):
    """time during which at least two intervals are open"""
    ev = []
    for s, e in iv:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    n, last, tot = 0, None, 0
    for t, d in ev:
        if n >= 2:
            tot += t - last
        n += d
        last = t
    return tot


rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'])
            for r in rows)
runs, cur = [], [iv[0]]
for x in iv[1:]:
    if x[0] - max(c[1] for c in cur[-50:]) > 30e6:
        runs.append(cur)
        cur = []
    cur.append(x)
runs.append(cur)
for r in runs:
    obj = [(s, e) for s, e, n in r if 'objective_kernel' in n]
    if len(obj) < 100:
        continue
    t0, t1 = min(s for s, e in obj), max(e for s, e in obj)
    allk = [(s, e) for s, e, n in r if s >= t0 and e <= t1]
    print('span %.3f s  some kernel %.3f  objective %.3f  two objective kernels %.3f  '
          'sum of objective durations %.3f  launches %d' % (
              (t1 - t0) / 1e9, union(allk) / 1e9, union(obj) / 1e9, overlap2(obj) / 1e9,
              sum(e - s for s, e in obj) / 1e9, len(obj)))
