#!/bin/bash
# one bench line per configuration of DESIGN.md section 5 -> gpurun_out/configs_<tag>.jsonl
tag=${1:-x}
cd $GRAFT_REPO_ROOT
out=gpurun_out/configs_$tag.jsonl
: > $out
run() { python bench.py "$@" 2>/dev/null | tail -1 >> $out; }
run --steps 3 --warmup 1
run --steps 3 --warmup 1 --per-spectrum-templates --no-cpu-baseline
run --ccf-every 9 --steps 2 --warmup 1 --cpu-sample 64
run --workload cfg2 --spectra 1000 --steps 5 --warmup 1 --cpu-sample 256
run --evaluator nn --steps 3 --warmup 1
run --refine --steps 2 --warmup 1 --no-cpu-baseline
run --resolution-matrix --steps 2 --warmup 1 --no-cpu-baseline
# (--process / --desi-file run the reference's default, second_minimizer = True;
# --process-no-bfgs is the Nelder-Mead-only add-on)
run --spectra 2000 --steps 1 --warmup 1 --cpu-sample 8 --process 2000 --process-cpu-sample 8
run --spectra 2000 --steps 1 --warmup 1 --no-cpu-baseline --process 2000 --process-no-bfgs
run --spectra 2000 --steps 1 --warmup 1 --no-cpu-baseline --desi-file 500
run --spectra 2000 --steps 1 --warmup 1 --no-cpu-baseline --desi-file 500 --process-no-bfgs
run --spectra 2000 --steps 1 --warmup 1 --no-cpu-baseline --desi-file 500 --desi-nfiles 16
run --spectra 2000 --steps 1 --warmup 1 --no-cpu-baseline --desi-file 500 --desi-nfiles 16 --process-no-bfgs
run --spectra 62500 --steps 2 --warmup 1 --no-cpu-baseline
# a library of realistic size (17 600 templates, dimensions of different length,
# 440 MB per arm: gathers served from HBM), the default step and the optimiser
run --grid 40,11,8,5 --steps 3 --warmup 1 --no-cpu-baseline
run --grid 40,11,8,5 --spectra 2000 --steps 1 --warmup 1 --no-cpu-baseline --process 2000
# the optimiser stage at full batch size, and on the NN evaluator
run --spectra 10000 --steps 1 --warmup 1 --no-cpu-baseline --process 10000
run --spectra 10000 --steps 1 --warmup 1 --no-cpu-baseline --process 10000 --process-no-bfgs
run --spectra 500 --steps 1 --warmup 1 --no-cpu-baseline --process 500
run --evaluator nn --spectra 2000 --steps 1 --warmup 1 --no-cpu-baseline --process 2000
run --evaluator nn --spectra 10000 --steps 1 --warmup 1 --no-cpu-baseline --process 10000
# the DESI driver on MLP libraries (random weights: the fits themselves mean nothing)
run --evaluator nn --spectra 2000 --steps 1 --warmup 1 --no-cpu-baseline --desi-file 500 --desi-nfiles 16
# round 4: the reference's own npoly (15), every spectrum on its own wavelength
# grid, the optimiser on the realistic-size library at full batch size
run --npoly 15 --steps 3 --warmup 1 --no-cpu-baseline
run --npoly 13 --steps 3 --warmup 1 --no-cpu-baseline
run --workload sdss --steps 3 --warmup 1 --cpu-sample 128
run --grid 40,11,8,5 --spectra 10000 --steps 1 --warmup 1 --no-cpu-baseline --process 10000
run --npoly 15 --spectra 2000 --steps 1 --warmup 1 --no-cpu-baseline --process 2000
# round 6: Delaunay libraries (find_simplex through the bucket grid; the rounds in rvs_nm_run)
run --evaluator tri --steps 3 --warmup 1 --cpu-sample 16
run --evaluator tri --spectra 2000 --steps 1 --warmup 1 --no-cpu-baseline --process 2000
python - <<PY
import json
for l in open("$out"):
    d = json.loads(l)
    c = d["config"]
    print(round(d["value"]), d["ms_per_step"], c["traffic_key"], d["roofline"]["frac"], d["roofline"]["traffic"],
          d["roofline_ccf"]["frac"], d["roofline_ccf"]["traffic"], (d.get("cpu_baseline") or {}).get("value"),
          (d["kernels"].get("template_polylinear") or {}).get("alg_GBps"), (d.get("process") or {}).get("spectra_per_s"),
          (d.get("desi_file") or {}).get("fibres_per_s"))
PY
