#!/bin/bash
# builds tools/perf/nn_bench.hip with variants of nn.hip (NN_VARIANTS: one word of
# comma-separated -D flags per variant, "base" = none) and prints the template
# stage's time at B rows (default 10000)
cd $GRAFT_REPO_ROOT
mkdir -p tools/perf/_bin gpurun_out
: > gpurun_out/nn_variants.log
for v in ${NN_VARIANTS:-base}; do
  flags=""
  [ "$v" != base ] && flags=$(echo $v | tr ',' ' ')
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-unused-value -Wno-unused-result $flags -I include \
    -o tools/perf/_bin/nn_bench_v tools/perf/nn_bench.hip 2>/dev/null || { echo "$v: build failed"; continue; }
  for b in ${NN_ROWS:-10000 440}; do
    echo "[$v] $(timeout 120 tools/perf/_bin/nn_bench_v $b 2>&1 | tail -1)" | tee -a gpurun_out/nn_variants.log
  done
done
