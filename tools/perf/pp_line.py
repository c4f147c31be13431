import json,sys
for l in sys.stdin:
    if l.startswith('{"metric"'):
        d=json.loads(l); p=d.get('process') or {}
        print(d['config'].get('workload','')[:30], 'value', d['value'], 'process', p.get('spectra_per_s'), p.get('stage_s'), 'evals', p.get('objective_evals'), 'parity', p.get('parity'))
