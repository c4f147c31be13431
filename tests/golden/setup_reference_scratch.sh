#!/bin/bash
# Scratch copy of the reference package under /tmp/oracle (never inside the repo),
# as SURVEY.md Appendix A: `_version.py` stub (normally written by setuptools_scm)
# and the cffi spline module compiled from the reference's own spliner.c under the
# conda 3.9 interpreter.  The make_golden*.py scripts import the reference from it.
set -e
rm -rf /tmp/oracle/rvspecfit
mkdir -p /tmp/oracle
cp -r /root/reference/py/rvspecfit /tmp/oracle/rvspecfit
echo "version = '0.0.probe'" > /tmp/oracle/rvspecfit/_version.py
cat > /tmp/oracle/build_spliner.py <<'PY'
import cffi
ffibuilder = cffi.FFI()
ffibuilder.set_source("rvspecfit._spliner", open('/root/reference/py/rvspecfit/src/spliner.c').read(), extra_compile_args=["-std=c99"])
ffibuilder.cdef("""
void construct(double *xs, double *ys, int N, double *A, double *B, double *C, double *D, double *h);
int evaler(double *evalx, int nevalx,  int N, double *xs, double *hs, double *As, double *Bs, double *Cs, double *Ds, int log_step, double *ret);
""")
ffibuilder.compile(verbose=False)
PY
cd /tmp/oracle && /opt/conda/bin/python3.9 -W ignore build_spliner.py
/opt/conda/bin/python3.9 -W ignore -c "
import sys, types
sys.path.insert(0, '/tmp/oracle'); sys.modules['numba'] = None
sys.modules['numdifftools'] = types.ModuleType('numdifftools')
from rvspecfit import spec_fit, fitter_ccf, make_ccf, vel_fit, spliner
print('reference importable from /tmp/oracle')"
