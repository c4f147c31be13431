#!/bin/bash
# tools/perf/clock_sample.sh <binary> [args...] -- run a timing binary in the
# background and sample the GPU's clock / power (rocm-smi) while it runs.
"$@" > /tmp/clk_run.log 2>&1 &
pid=$!
sleep 3
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|fclk\|mclk\|power\|junction" | tr '\n' ' '
  echo
  sleep 0.5
done
wait $pid
tail -2 /tmp/clk_run.log
