import json,sys
for l in sys.stdin:
    if l.startswith('{"metric"'):
        d=json.loads(l); p=d.get('process') or {}
        if p:
            print({k:p.get(k) for k in ('spectra','spectra_per_s','seconds','streams','single_stream_seconds','stage_s','nm_rounds','nm_iterations_mean','nm_iterations_max')})
        f=d.get('desi_file') or {}
        if f:
            print({k:f.get(k) for k in ('fibres','files','fibres_per_s','seconds','stage_s')})
