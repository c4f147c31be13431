"""consecutive kernels of one queue late in the last two-stream process call of a
rocprofv3 kernel trace: name, start relative to the first, duration, gap before (us)"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
obj = [r for r in rows if 'objective_kernel' in r['Kernel_Name']]
obj.sort(key=lambda r: int(r['Start_Timestamp']))
# calls split at 50 ms gaps; take the last call with two queues
calls, cur = [], [obj[0]]
for a, b in zip(obj, obj[1:]):
    if int(b['Start_Timestamp']) - int(a['End_Timestamp']) > 50e6:
        calls.append(cur)
        cur = []
    cur.append(b)
calls.append(cur)
two = [c for c in calls if len({r['Queue_Id'] for r in c}) == 2 and len(c) > 1000][-1]
t_end = int(two[-1]['End_Timestamp'])
q = two[-1]['Queue_Id']
lo = t_end - int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else t_end - 3000000
sel = [r for r in rows if r['Queue_Id'] == q and lo <= int(r['Start_Timestamp']) <= t_end]
sel.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(sel[0]['Start_Timestamp'])
prev = None
for r in sel[:60]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%-34s start %8.1f dur %6.1f gap %6.1f grid %s' % (
        r['Kernel_Name'].split('(')[0][-34:], (s - t0) / 1e3, (e - s) / 1e3,
        0.0 if prev is None else (s - prev) / 1e3, r['Grid_Size_X']))
    prev = e
