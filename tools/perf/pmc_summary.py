import sys, csv, glob, collections
d = sys.argv[1]
f = glob.glob(d + '/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][:40]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in agg.items():
    if not any(x in k for x in ('chisq', 'ccf', 'spline', 'polylin', 'vsini', 'nn_')): continue
    print(k, {n: '%.4g (n=%d)' % (sum(v) / len(v), len(v)) for n, v in c.items()})
