#!/bin/bash
# tools/perf/trace_chain.sh <S> [bench flags]: kernel trace of bench.py --process S,
# reduced by trace_chain.py (durations and idle time per kernel of the chain)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
S=${1:-500}; shift
rm -rf /tmp/tc && mkdir -p /tmp/tc
timeout 600 rocprofv3 --kernel-trace -d /tmp/tc -o p --output-format csv -- python3 bench.py --spectra $S --process $S --steps 1 --warmup 1 --no-cpu-baseline "$@" > /tmp/tc/bench.log 2>&1
tail -1 /tmp/tc/bench.log | python3 tools/perf/proc_line.py 2>/dev/null | tail -2
f=$(find /tmp/tc -name 'p_kernel_trace.csv' | head -1)
python3 tools/perf/trace_chain.py $f
