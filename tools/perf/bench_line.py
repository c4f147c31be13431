"""a few fields of a bench.py JSON line (stdin)"""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['unit'], d['ms_per_step'], 'ms; roofline', d['roofline']['frac'],
      'ccf', d['roofline_ccf']['frac'], d['roofline_ccf']['counter_backed'],
      'cpu', d.get('cpu_baseline'))
