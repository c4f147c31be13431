#!/bin/bash
# tools/perf/profile_r03.sh <tag>: everything the round's bench lines cite, in the
# order that lets a line carry the counters measured on ITS configuration:
#   1. per-configuration HBM traffic (pmc_traffic.sh)  -> profiles/r03_pmc_traffic.json
#   2. SQ counters of the default configuration        -> profiles/r03_sq_counters.json
#   3. default line + rocprofv3 kernel stats (+ raw FETCH / WRITE sums)
#   4. one line per configuration (all_configs.sh)
#   5. the optimiser stage: kernel stats of `--process 10000`, phase budget of the
#      objective kernel
tag=${1:-r03_x}
R=$GRAFT_REPO_ROOT
cd $R
bash tools/perf/pmc_traffic.sh $tag > gpurun_out/pmc_traffic_$tag.log 2>&1
cp gpurun_out/pmc_traffic_$tag.json profiles/r03_pmc_traffic.json
XC_ARGS="" bash tools/perf/xc_counters.sh $tag > gpurun_out/xc_counters_$tag.log 2>&1
cp gpurun_out/xc_counters_$tag.json profiles/r03_sq_counters.json
bash tools/perf/profile_round.sh $tag > gpurun_out/profile_round_$tag.log 2>&1
bash tools/perf/all_configs.sh $tag > gpurun_out/all_configs_$tag.log 2>&1
bash tools/perf/prof_cmd.sh ${tag}_process10k --spectra 10000 --steps 1 --warmup 1 --no-cpu-baseline --process 10000 > gpurun_out/prof_process_$tag.log 2>&1
bash tools/perf/prof_cmd.sh ${tag}_grid_big --grid 40,11,8,5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_gridbig_$tag.log 2>&1
bash tools/perf/obj_phases.sh > gpurun_out/obj_phases_$tag.log 2>&1
tail -3 gpurun_out/all_configs_$tag.log
tail -2 gpurun_out/obj_phases_$tag.log
tail -c 400 gpurun_out/bench_$tag.json
