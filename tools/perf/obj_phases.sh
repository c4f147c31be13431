#!/bin/bash
# per-phase clock budget of the fused objective kernel (debug build, -DRVS_OBJ_TIMING)
cd $GRAFT_REPO_ROOT
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DRVS_OBJ_TIMING -c rvspecfit_amd/csrc/objective.hip -o /tmp/objective_t.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/librvsgpu_t.so /tmp/objective_t.o $(ls rvspecfit_amd/csrc/_build/*.o | grep -v objective.o)
cp rvspecfit_amd/librvsgpu.so /tmp/librvsgpu_orig.so
cp /tmp/librvsgpu_t.so rvspecfit_amd/librvsgpu.so
python - <<'PY'
import ctypes, sys, numpy as np, torch, runpy
sys.argv = ['x', '1000']
src = open('tools/perf/proc_time.py').read().replace('for it in range(2):', 'for it in range(1):')
exec(compile(src, 'p', 'exec'))
L = _lib.lib()
buf = (ctypes.c_ulonglong * 24)()
L.rvs_dbg_read.argtypes = [ctypes.c_void_p]
L.rvs_dbg_read(ctypes.addressof(buf))
t = np.array(buf[:10], dtype=float)
names = ['locate', 'gather+exp', 'vsini', 'spline', 'tv+normal', 'cholesky+solve', 'resid', 'model pass', 'wave reduce', 'fold']
print({n: round(float(v / t.sum()), 3) for n, v in zip(names, t)}, 'total ticks', t.sum())
PY
cp /tmp/librvsgpu_orig.so rvspecfit_amd/librvsgpu.so
