#!/bin/bash
# per-phase clock budget of ccf_xcorr_kernel (debug build, -DRVS_XC_TIMING)
cd $GRAFT_REPO_ROOT
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DRVS_XC_TIMING -c rvspecfit_amd/csrc/ccf_fft.hip -o /tmp/ccf_fft_t.o 2>/dev/null
make -C rvspecfit_amd/csrc -j8 > /dev/null 2>&1
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/librvsgpu_t.so /tmp/ccf_fft_t.o $(ls rvspecfit_amd/csrc/_build/*.o | grep -v /ccf_fft.o)
cp rvspecfit_amd/librvsgpu.so /tmp/librvsgpu_orig.so
cp /tmp/librvsgpu_t.so rvspecfit_amd/librvsgpu.so
python - <<'PY'
import ctypes, sys, numpy as np
sys.argv = ['bench.py', '--spectra', '2000', '--steps', '2', '--warmup', '1', '--no-cpu-baseline']
import bench
bench.main()
from rvspecfit_amd import _lib
L = _lib.lib()
buf = (ctypes.c_ulonglong * 16)()
L.rvs_dbg_read_xc.argtypes = [ctypes.c_void_p]
L.rvs_dbg_read_xc(ctypes.addressof(buf))
t = np.array(buf[:3], dtype=float)
names = ['operands', 'fft passes', 'read-back']
print({n: round(float(v / t.sum()), 3) for n, v in zip(names, t)}, 'total ticks', t.sum())
PY
cp /tmp/librvsgpu_orig.so rvspecfit_amd/librvsgpu.so
