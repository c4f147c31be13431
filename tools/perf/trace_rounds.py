"""Round structure of vel_fit.process from a rocprofv3 kernel trace (p_kernel_trace.csv):
for the LAST process call of the run (bench.py --process: the single-stream repeat) and
the one before it (the two-stream call), the time with 0 / 1 / 2 objective kernels
running, and -- per queue -- the gap between the end of one objective kernel and the
start of the next (the chain of small kernels in between), by launch size."""
import csv
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    obj = [r for r in rows if 'objective_kernel' in r['Kernel_Name']]
    obj.sort(key=lambda r: int(r['Start_Timestamp']))
    # split into calls at gaps > 50 ms between objective kernels
    calls, cur = [], [obj[0]]
    for a, b in zip(obj, obj[1:]):
        if int(b['Start_Timestamp']) - int(a['End_Timestamp']) > 50e6:
            calls.append(cur)
            cur = []
        cur.append(b)
    calls.append(cur)
    print('process calls (objective launches each):', [len(c) for c in calls])
    for ci, c in enumerate(calls):
        if len(c) < 1000:
            continue
        t0, t1 = int(c[0]['Start_Timestamp']), int(c[-1]['End_Timestamp'])
        ev = []
        for r in c:
            ev.append((int(r['Start_Timestamp']), 1))
            ev.append((int(r['End_Timestamp']), -1))
        ev.sort()
        lvl, last, tl = 0, t0, defaultdict(int)
        for t, d in ev:
            tl[lvl] += t - last
            last = t
            lvl += d
        queues = sorted({r['Queue_Id'] for r in c})
        print('call %d: %.3f s, queues %s; objective kernels running: '
              % (ci, (t1 - t0) / 1e9, queues) +
              ', '.join('%d: %.3f s' % (k, v / 1e9) for k, v in sorted(tl.items())))
        for q in queues:
            cq = [r for r in c if r['Queue_Id'] == q]
            bins = defaultdict(list)
            for a, b in zip(cq, cq[1:]):
                gap = int(b['Start_Timestamp']) - int(a['End_Timestamp'])
                g = int(b['Grid_Size_X']) // int(b['Workgroup_Size_X'])
                key = 0 if g < 64 else 1 if g < 256 else 2 if g < 1024 else 3
                bins[key].append((gap, int(b['End_Timestamp']) - int(b['Start_Timestamp'])))
            for k in sorted(bins):
                v = bins[k]
                print('  queue %s jobs %s: %5d launches, gap before mean %.1f us '
                      '(sum %.3f s), duration mean %.1f us (sum %.3f s)' %
                      (q, ['<64', '64-255', '256-1023', '>=1024'][k], len(v),
                       sum(x[0] for x in v) / len(v) / 1e3, sum(x[0] for x in v) / 1e9,
                       sum(x[1] for x in v) / len(v) / 1e3, sum(x[1] for x in v) / 1e9))


if __name__ == '__main__':
    main()
