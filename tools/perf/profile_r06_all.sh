#!/bin/bash
# tools/perf/profile_r06_all.sh <tag> <part>: everything the round-6 bench lines
# cite.  part 1: per-configuration HBM traffic (-> profiles/r06_pmc_traffic.json),
# SQ counters of the default configuration (-> profiles/r06_sq_counters.json),
# counters of the objective kernel on both library sizes (->
# profiles/r06_obj_counters.json).  part 2 (reads those files): default line +
# kernel stats, one line per configuration, kernel stats of --process 10000 and of
# the 17 600-template library, phase budget of the objective kernel.
tag=${1:-r06_x}; part=${2:-1}
R=$GRAFT_REPO_ROOT
cd $R
if [ "$part" = 1 ]; then
  bash tools/perf/pmc_traffic.sh $tag > gpurun_out/pmc_traffic_$tag.log 2>&1
  cp gpurun_out/pmc_traffic_$tag.json profiles/r06_pmc_traffic.json
  XC_ARGS="" bash tools/perf/xc_counters.sh $tag > gpurun_out/xc_counters_$tag.log 2>&1
  cp gpurun_out/xc_counters_$tag.json profiles/r06_sq_counters.json
  export OBJ_BENCH_SKIP_REF=1
  for g in 7,7,7,7 40,11,8,5; do
    bash tools/perf/obj_counters.sh ${tag}_${g//,/x} 9000 2 $g > gpurun_out/objc_${tag}_${g//,/x}.log 2>&1
  done
  unset OBJ_BENCH_SKIP_REF
  python3 - <<PY
import json
out = {}
for g in ('7x7x7x7', '40x11x8x5'):
    d = json.load(open('gpurun_out/obj_counters_${tag}_%s.json' % g))
    k = [v for n, v in d['kernels'].items() if 'objective_kernel' in n][0]
    alg = 9000 * sum(16 * n * 4 for n in (6215, 5303, 6449))
    out[g] = dict(
        what='objective_kernel<10,false,false>, 9000 jobs x 3 DESI arms at random in-grid '
             'parameters, jobs in cell order (tools/perf/obj_bench, tools/perf/obj_counters.sh)',
        avg_duration_ms=round(k['avg_duration_ns'] / 1e6, 3),
        us_per_arm_evaluation_per_cu=round(k['avg_duration_ns'] / 1e3 * 256 / 27000, 2),
        valu_busy=k.get('valu_busy'), lds_busy=k.get('lds_busy'), l2_hit_rate=k.get('l2_hit_rate'),
        valu_instructions_per_block=round(k['SQ_INSTS_VALU'] / 27000, 0),
        lds_instructions_per_block=round(k['SQ_INSTS_LDS'] / 27000, 0),
        lds_bank_conflict_frac=round(k['SQ_LDS_BANK_CONFLICT'] / max(1.0, k['SQ_LDS_IDX_ACTIVE']), 3),
        wait_inst_frac_of_wave_cycles=round(k['SQ_WAIT_INST_ANY'] / k['SQ_WAVE_CYCLES'], 3),
        fetch_GB_per_launch=round(2 * k['FETCH_SIZE'] * 1024 / 1e9, 2),
        write_GB_per_launch=round(k['WRITE_SIZE'] * 1024 / 1e9, 3),
        gathered_GB_per_launch_algorithmic=round(alg / 1e9, 2))
json.dump(out, open('gpurun_out/obj_counters_merged_${tag}.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
  cp gpurun_out/obj_counters_merged_$tag.json profiles/r06_obj_counters.json
else
  for f in pmc_traffic xc_counters obj_counters_merged; do
    [ -f gpurun_out/${f}_$tag.json ] || echo "missing gpurun_out/${f}_$tag.json (run part 1 first; copy it to profiles/)"
  done
  bash tools/perf/profile_round.sh $tag > gpurun_out/profile_round_$tag.log 2>&1
  bash tools/perf/all_configs.sh $tag > gpurun_out/all_configs_$tag.log 2>&1
  bash tools/perf/prof_cmd.sh ${tag}_process10k --spectra 10000 --steps 1 --warmup 1 --no-cpu-baseline --process 10000 > gpurun_out/prof_process_$tag.log 2>&1
  bash tools/perf/prof_cmd.sh ${tag}_grid_big --grid 40,11,8,5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_gridbig_$tag.log 2>&1
  bash tools/perf/prof_cmd.sh ${tag}_npoly15 --npoly 15 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_npoly15_$tag.log 2>&1
  bash tools/perf/obj_phases.sh > gpurun_out/obj_phases_$tag.log 2>&1
  tail -30 gpurun_out/all_configs_$tag.log
  tail -2 gpurun_out/obj_phases_$tag.log
  tail -c 400 gpurun_out/bench_$tag.json
fi
