#!/bin/bash
# per-phase clock budget of ccf_preprocess_kernel (debug build, -DRVS_PP_TIMING)
cd $GRAFT_REPO_ROOT
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DRVS_PP_TIMING -c rvspecfit_amd/csrc/ccf.hip -o /tmp/ccf_t.o 2>/dev/null
make -C rvspecfit_amd/csrc -j8 > /dev/null 2>&1
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/librvsgpu_t.so /tmp/ccf_t.o $(ls rvspecfit_amd/csrc/_build/*.o | grep -v /ccf.o)
cp rvspecfit_amd/librvsgpu.so /tmp/librvsgpu_orig.so
cp /tmp/librvsgpu_t.so rvspecfit_amd/librvsgpu.so
python - <<'PY'
import ctypes, subprocess, sys, json, numpy as np
sys.argv = ['bench.py', '--spectra', '2000', '--steps', '2', '--warmup', '1', '--no-cpu-baseline']
import bench
bench.main()
from rvspecfit_amd import _lib
L = _lib.lib()
buf = (ctypes.c_ulonglong * 16)()
L.rvs_dbg_read_pp.argtypes = [ctypes.c_void_p]
L.rvs_dbg_read_pp(ctypes.addressof(buf))
t = np.array(buf[:12], dtype=float)
names = ['load+errmedian+medfilt', 'gapfill', 'median sort', 'binned sort', 'LM tail', 'normalise', 'rebin', 'LM: setup+eval0', 'LM: normal eq', 'LM: band solve', 'LM: trial eval', 'LM: accept+eval']
print({n: round(float(v / t.sum()), 3) for n, v in zip(names, t)}, 'total ticks', t.sum())
PY
cp /tmp/librvsgpu_orig.so rvspecfit_amd/librvsgpu.so
