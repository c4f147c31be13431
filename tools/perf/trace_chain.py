"""Where the wall time of vel_fit.process goes, from a rocprofv3 kernel trace
(p_kernel_trace.csv of `bench.py --process N`): per queue, between the first and the
last objective kernel, every kernel's summed duration and the summed idle time in front
of it (its start minus the end of the kernel before it on the same queue).
usage: trace_chain.py <p_kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    for r in rows:
        r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    obj = [r for r in rows if 'objective_kernel' in r['Kernel_Name']]
    t0, t1 = min(r['s'] for r in obj), max(r['e'] for r in obj)
    print('objective kernels from %.3f s to %.3f s of the trace: %.3f s'
          % (0, (t1 - t0) / 1e9, (t1 - t0) / 1e9))
    for q in sorted({r['Queue_Id'] for r in obj}):
        rq = sorted((r for r in rows if r['Queue_Id'] == q and t0 <= r['s'] <= t1),
                    key=lambda r: r['s'])
        agg = defaultdict(lambda: [0, 0, 0])
        durs = defaultdict(list)
        for a, b in zip(rq, rq[1:]):
            name = b['Kernel_Name'].split('(')[0].replace('void ', '')[:48]
            g = agg[name]
            g[0] += 1
            g[1] += b['e'] - b['s']
            g[2] += max(0, b['s'] - a['e'])
            durs[name].append(b['e'] - b['s'])
        tot_d = sum(v[1] for v in agg.values())
        tot_g = sum(v[2] for v in agg.values())
        print('queue %s: %d kernels, busy %.3f s, idle in front of kernels %.3f s'
              % (q, len(rq), tot_d / 1e9, tot_g / 1e9))
        for name, (n, d, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:16]:
            v = sorted(durs[name])
            print('  %-48s %6d x  run %8.1f ms (%.1f us; p10 %.1f p50 %.1f p90 %.1f)  '
                  'idle before %7.1f ms (%.1f us)'
                  % (name, n, d / 1e6, d / n / 1e3, v[len(v) // 10] / 1e3,
                     v[len(v) // 2] / 1e3, v[(9 * len(v)) // 10] / 1e3, g / 1e6,
                     g / n / 1e3))


if __name__ == '__main__':
    main()
