"""GPU busy time inside the span of a kernel's launches, from a rocprofv3 kernel trace:
usage trace_busy.py p_kernel_trace.csv [kernel-name substring]"""
import csv
import sys


def union(iv, a, b):
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if e < a or s > b:
            continue
        s, e = max(s, a), min(e, b)
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    pat = sys.argv[2] if len(sys.argv) > 2 else 'objective_kernel'
    sel = [r for r in rows if pat in r['Kernel_Name']]
    iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
    ivs = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in sel)
    # the longest run of launches without a gap of more than 0.2 s (one process call)
    t0, t1 = ivs[0][0], ivs[-1][1]
    print('%d kernels, %d of %s' % (len(rows), len(sel), pat))
    print('span %.3f s; some kernel running %.3f s; a %s running %.3f s; '
          'sum of its durations %.3f s' %
          ((t1 - t0) / 1e9, union(iv, t0, t1) / 1e9, pat, union(ivs, t0, t1) / 1e9,
           sum(e - s for s, e in ivs) / 1e9))


if __name__ == '__main__':
    main()
