"""vel_fit.process on S bench spectra with the rounds as bookkeeping kernels and as the
chain of stand-alone kernels (RVS_NM_GLUE=0): bit-equality of the results, and how
many shrink steps (scipy's rare fifth branch, run by the host between two windows)
the run contained."""
import os
import sys
S = sys.argv[1] if len(sys.argv) > 1 else '2000'
sys.argv = [sys.argv[0], S]
src = open('tools/perf/proc_time.py').read().replace('for it in range(2):', 'for it in range(1):')
out = {}
for glue in ('1', '0'):
    from rvspecfit_amd import _lib
    _lib.set_option('nm_glue', int(glue))
    g = {}
    exec(compile(src, 'p', 'exec'), g)
    out[glue] = g['r']
a, b = out['1'], out['0']
import torch
print('nit equal', torch.equal(a['nm_nit'], b['nm_nit']), 'nfev equal',
      torch.equal(a['nm_nfev'], b['nm_nfev']), 'vel equal', torch.equal(a['vel'], b['vel']),
      'chisq equal', torch.equal(a['chisq'], b['chisq']))
# a shrink adds N function values and one iteration: nfev - (N + 1) - (1..2) * nit
nit, nfev = a['nm_nit'].double(), a['nm_nfev'].double()
print('simplices with nfev > 2 nit + 8 (at least one shrink):',
      int((nfev > 2 * nit + 8).sum().item()), 'of', nit.numel())
