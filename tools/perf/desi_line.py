import json, sys
for l in sys.stdin:
    if l.startswith('{"metric"'):
        f = json.loads(l).get('desi_file') or {}
        print({k: f.get(k) for k in ('fibres', 'files', 'files_per_batch', 'workers', 'fibres_per_s', 'seconds', 'stage_s')})
