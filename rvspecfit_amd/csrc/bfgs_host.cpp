// bfgs_host.cpp -- host side of the second minimiser of vel_fit.process
// (vel_fit.py:653-658: scipy.optimize.minimize(method='BFGS', hess_inv0=...)).
//
// S independent BFGS runs advance in lock-step: every run is a resumable state
// machine (bfgs_machine.h: scipy's _minimize_bfgs with its line searches, statement
// for statement) that returns whenever it needs objective values (one point, the n
// forward-difference points of a gradient, or both); the driver gathers the requests
// of all runs into ONE batch for the caller's objective and resumes them with the
// values.  The same machine runs in a kernel, one thread per spectrum, for the
// libraries whose objective the library can launch itself (bfgs_dev.hip:
// rvs_bfgs_run); this host driver serves every other objective (a Python callable:
// Delaunay evaluators, resolution matrices) and is what the CPU suite pins to the
// scipy-pinned Python restatement (tests/refmachines/bfgs_scipy_restated.py,
// tests/test_tools_cpu.py).  It exists because Python generators cost ~20 us per
// resume: 2.6 s per 2000 spectra, more than the Nelder-Mead stage on the GPU.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/rvsgpu.h"
#include "bfgs_machine.h"

namespace {

using rvs_bfgs::MAXN;
using rvs_bfgs::Run;

struct Driver {
  int S, n;
  std::vector<Run> runs;
  std::vector<int> order;  // spectra with a pending request, ascending
  int64_t rounds = 0;
};

}  // namespace

extern "C" {

void *rvs_bfgs_begin(int S, int n, const double *x0, const double *hess_inv0,
                     double gtol, double c1, double c2, double xrtol,
                     int maxiter) {
  if (S < 1 || n < 1 || n > MAXN || !x0) return nullptr;
  Driver *d = new Driver;
  d->S = S;
  d->n = n;
  d->runs.resize(S);
  for (int s = 0; s < S; s++) {
    Run &r = d->runs[s];
    rvs_bfgs::init(r, n, x0 + (int64_t)s * n, hess_inv0, gtol, c1, c2, xrtol,
                   maxiter);
    rvs_bfgs::advance(r);  // runs to the first request (f and g at x0)
  }
  return d;
}

int64_t rvs_bfgs_pending(void *h, int64_t *idx, double *X, int64_t cap_rows) {
  Driver *d = static_cast<Driver *>(h);
  if (!d) return -1;
  d->order.clear();
  int64_t rows = 0;
  const int n = d->n;
  for (int s = 0; s < d->S; s++) {
    Run &r = d->runs[s];
    if (r.done || r.nrows == 0) continue;
    if (rows + r.nrows > cap_rows) return -2;
    for (int q = 0; q < r.nrows; q++) {
      idx[rows + q] = s;
      std::memcpy(X + (rows + q) * n, r.rows + q * n, sizeof(double) * n);
    }
    rows += r.nrows;
    d->order.push_back(s);
  }
  return rows;
}

int rvs_bfgs_feed(void *h, const double *F, int64_t nrows) {
  Driver *d = static_cast<Driver *>(h);
  if (!d) return RVS_E_ARG;
  int64_t at = 0;
  for (int s : d->order) {
    Run &r = d->runs[s];
    if (at + r.nrows > nrows) return RVS_E_ARG;
    const int m = r.nrows;
    for (int q = 0; q < m; q++) r.vals[q] = F[at + q];
    at += m;
    rvs_bfgs::advance(r);
  }
  if (at != nrows) return RVS_E_ARG;
  d->rounds += 1;
  d->order.clear();
  return 0;
}

int rvs_bfgs_result(void *h, double *x, double *fun, int32_t *nit,
                    int32_t *nfev, int32_t *status, double *hess_inv,
                    int64_t *rounds) {
  Driver *d = static_cast<Driver *>(h);
  if (!d) return RVS_E_ARG;
  const int n = d->n;
  for (int s = 0; s < d->S; s++) {
    const Run &r = d->runs[s];
    if (!r.done) return RVS_E_ARG;
    std::memcpy(x + (int64_t)s * n, r.xk, sizeof(double) * n);
    fun[s] = r.fval;
    nit[s] = r.nit;
    nfev[s] = r.nfev;
    status[s] = r.status;
    if (hess_inv)
      std::memcpy(hess_inv + (int64_t)s * n * n, r.Hk, sizeof(double) * n * n);
  }
  if (rounds) *rounds = d->rounds;
  return 0;
}

void rvs_bfgs_end(void *h) {
  Driver *d = static_cast<Driver *>(h);
  if (!d) return;
  delete d;
}

}  // extern "C"
