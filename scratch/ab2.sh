#!/bin/bash
for v in $@; do
  echo "RVS_XCORR_TPER=$v"
  RVS_XCORR_TPER=$v python bench.py --spectra 2000 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['kernels'])"
done
