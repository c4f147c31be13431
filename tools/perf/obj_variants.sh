#!/bin/bash
# builds librvsgpu with variants of the objective kernel in turn (OBJ_VARIANTS:
# hipcc -D flags, one variant per word; default: block sizes 512 / 768) and prints
# the optimiser-stage rate (bench.py --process)
cd $GRAFT_REPO_ROOT
cp rvspecfit_amd/librvsgpu.so /tmp/librvsgpu_orig.so
trap 'cp /tmp/librvsgpu_orig.so rvspecfit_amd/librvsgpu.so' EXIT
for v in ${OBJ_VARIANTS:--DOBJ_NT=512 -DOBJ_NT=768}; do
  hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 $v -c rvspecfit_amd/csrc/objective.hip -o /tmp/obj_v.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o rvspecfit_amd/librvsgpu.so /tmp/obj_v.o $(ls rvspecfit_amd/csrc/_build/*.o | grep -v /objective.o)
  python -m pytest tests/test_gpu_parity.py -x -q -k "objective_fused or process_golden" 2>&1 | tail -1
  python bench.py --spectra ${OBJ_S:-4000} --steps 1 --warmup 1 --no-cpu-baseline --process ${OBJ_S:-4000} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['process']['spectra_per_s'], d['process']['stage_s'])"
done
