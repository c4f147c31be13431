#!/bin/bash
# round 4, final additions (after r04_e): the lines the late changes moved
cd $GRAFT_REPO_ROOT
out=gpurun_out/configs_r04_f.jsonl
: > $out
run() { python bench.py "$@" 2>/dev/null | tail -1 >> $out; }
run --steps 3 --warmup 1
run --workload sdss --steps 3 --warmup 1 --cpu-sample 128
run --workload sdss --spectra 10000 --steps 1 --warmup 1 --no-cpu-baseline --process 10000
run --workload sdss --spectra 2000 --steps 1 --warmup 1 --cpu-sample 8 --process 2000 --process-cpu-sample 8
run --npoly 15 --spectra 2000 --steps 1 --warmup 1 --no-cpu-baseline --process 2000
run --npoly 15 --spectra 10000 --steps 1 --warmup 1 --no-cpu-baseline --process 10000
run --npoly 15 --steps 3 --warmup 1 --no-cpu-baseline
python tools/perf/sdss_setup_time.py 10000 1 > gpurun_out/sdss_setup_r04_f.txt 2>/dev/null
python tools/perf/sdss_setup_time.py 2000 0 >> gpurun_out/sdss_setup_r04_f.txt 2>/dev/null
cat gpurun_out/sdss_setup_r04_f.txt
python - <<PY
import json
for l in open("$out"):
    d = json.loads(l)
    p = d.get("process") or {}
    print(round(d["value"]), d["ms_per_step"], d["config"]["traffic_key"][:60], d["roofline"]["frac"], p.get("spectra_per_s"), (p.get("roofline") or {}).get("frac"), (p.get("parity") or {}))
PY
