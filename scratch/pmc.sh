#!/bin/bash
# usage: scratch/pmc.sh <tag> <counters...>
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --spectra 2000 --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/*/ | head
