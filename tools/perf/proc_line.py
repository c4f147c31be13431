"""one-line summary of the `process` object of a bench.py --process JSON line (stdin)"""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
p = d['process']
print(' '.join(sys.argv[1:]), p['spectra_per_s'], 'spectra/s', p['seconds'], 's;',
      p['stage_s'], 'single stream', p['single_stream_seconds'], 's; nit',
      p['nm_iterations_mean'], 'evals', p['objective_evals'], 'rows launched',
      p.get('nm_launched_rows'), 'frac', p['roofline']['frac'])
