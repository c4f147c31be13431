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
# (indices 18..23: wave 0 inside 'cholesky+solve', clocked without barriers)
sub = np.array(buf[18:24], dtype=float)
tot = t.sum() + sub.sum()
print({n: round(float(v / tot), 3) for n, v in zip(names, t)}, 'total ticks', tot)
print('all 24 clocks / total:', [round(float(v) / float(sum(buf[:24])), 3) for v in buf[:24]])
print({n: round(float(v / tot), 3) for n, v in zip(
    ['chol: rows from LDS', 'chol: factor', 'chol: row requests', 'chol: log',
     'chol: forward', 'chol: backward+det'], sub)})
PY
cp /tmp/librvsgpu_orig.so rvspecfit_amd/librvsgpu.so
