#!/bin/bash
# builds tools/perf/obj_bench.hip once per variant (OBJ_VARIANTS: one word of
# comma-separated -D flags per variant, "base" = none) and runs each with the
# argument sets in OBJ_ARGS (semicolon separated; default "3000 5")
cd $GRAFT_REPO_ROOT
mkdir -p tools/perf/_bin gpurun_out
: > gpurun_out/obj_bench.log
for v in ${OBJ_VARIANTS:-base}; do
  flags=""
  [ "$v" != base ] && flags=$(echo $v | tr ',' ' ')
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-unused-value -Wno-unused-result -DOBJ_ONLY_P=${OBJ_P:-10} $flags \
    -o tools/perf/_bin/obj_bench_v tools/perf/obj_bench.hip \
    -Lrvspecfit_amd -l:librvsgpu.so -Wl,-rpath,$GRAFT_REPO_ROOT/rvspecfit_amd 2>/dev/null || { echo "$v: build failed"; continue; }
  IFS=';' read -ra sets <<< "${OBJ_ARGS:-3000 5}"
  for a in "${sets[@]}"; do
    echo "[$v] $(timeout 90 tools/perf/_bin/obj_bench_v $a 2>&1 | tail -${OBJ_TAIL:-12})" | tee -a gpurun_out/obj_bench.log
  done
done
