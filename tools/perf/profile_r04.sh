#!/bin/bash
# round-4 measurement set: tools/perf/profile_r04.sh <tag> [what ...]
#   npoly     default step at --npoly 15 (+ kernel stats of the same command)
#   objc      SQ / TCC counters of the objective kernel on both library sizes
#   proc      bench.py --process 10000 (7^4 and 40x11x8x5), NN --process 10000
cd $GRAFT_REPO_ROOT
tag=$1; shift
what="${@:-npoly objc proc}"
mkdir -p gpurun_out
for w in $what; do
case $w in
npoly)
  python bench.py --npoly 15 --steps 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_npoly15_bench.json
  python bench.py --npoly 13 --steps 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_npoly13_bench.json
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_np && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_np -o p -- python3 $GRAFT_REPO_ROOT/bench.py --npoly 15 --steps 5 --no-cpu-baseline > /tmp/prof_np.log 2>&1; cp $(find /tmp/prof_np -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/${tag}_npoly15_kernel_stats.csv)
  python - <<PY
import json
for n in (13, 15):
    d = json.load(open('gpurun_out/${tag}_npoly%d_bench.json' % n))
    print('npoly', n, d['value'], 'spectra/s  roofline', d['roofline']['frac'], d['roofline']['achieved'], 'TF', d['roofline']['avg_launch_ms'], 'ms')
PY
  ;;
objc)
  export OBJ_BENCH_SKIP_REF=1
  for g in 7,7,7,7 40,11,8,5; do
    tools/perf/obj_counters.sh ${tag}_${g//,/x} 9000 2 $g > gpurun_out/${tag}_objc_${g//,/x}.log 2>&1
  done
  unset OBJ_BENCH_SKIP_REF
  ;;
proc)
  python bench.py --spectra 10000 --steps 1 --warmup 1 --no-cpu-baseline --process 10000 2>/dev/null | tail -1 > gpurun_out/${tag}_process10k.json
  python bench.py --spectra 10000 --steps 1 --warmup 1 --no-cpu-baseline --process 10000 --grid 40,11,8,5 2>/dev/null | tail -1 > gpurun_out/${tag}_process10k_grid40.json
  python bench.py --spectra 10000 --steps 1 --warmup 1 --no-cpu-baseline --process 10000 --evaluator nn 2>/dev/null | tail -1 > gpurun_out/${tag}_process10k_nn.json
  python - <<PY
import json
for n in ('', '_grid40', '_nn'):
    try:
        d = json.load(open('gpurun_out/${tag}_process10k%s.json' % n))['process']
        print('process10k' + n, d['spectra_per_s'], d['stage_s'], {k: d['roofline'][k] for k in ('achieved', 'frac', 'evaluations_per_s', 'gather_GBps', 'us_per_arm_evaluation_per_cu')})
    except Exception as e:
        print('process10k' + n, 'failed', e)
PY
  ;;
esac
done
