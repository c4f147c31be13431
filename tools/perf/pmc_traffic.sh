#!/bin/bash
# tools/perf/pmc_traffic.sh <tag>: HBM traffic per launch of the two big kernels
# for EVERY configuration bench.py reports, each measured on its own workload:
# two rocprofv3 counter passes per configuration (FETCH_SIZE, WRITE_SIZE; kernel
# trace only, the program directly after `--`), reduced by pmc_traffic.py into
# gpurun_out/pmc_traffic_<tag>.json keyed by the line's config.traffic_key.
# Copy the result to profiles/r04_pmc_traffic.json (bench.py reads it).
tag=${1:-x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/pmc_traffic_$tag.json
rm -f $out
cfg() {
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- \
      python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" \
      > /tmp/pmc_$c.log 2>&1
  done
  python3 $R/tools/perf/pmc_traffic.py $out /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE \
    /tmp/pmc_FETCH_SIZE.log "$tag" "$*"
}
cfg
cfg --per-spectrum-templates
cfg --ccf-every 9
cfg --workload cfg2 --spectra 1000
cfg --evaluator nn
cfg --refine
cfg --resolution-matrix
cfg --spectra 62500
cfg --grid 40,11,8,5
cfg --npoly 15
cfg --workload sdss
cfg --evaluator tri
cat $out | head -c 3000
