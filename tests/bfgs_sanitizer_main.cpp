#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
extern "C" {
void *rvs_bfgs_begin(int S, int n, const double *x0, const double *hess_inv0, double gtol, double c1, double c2, double xrtol, int maxiter);
int64_t rvs_bfgs_pending(void *h, int64_t *idx, double *X, int64_t cap_rows);
int rvs_bfgs_feed(void *h, const double *F, int64_t nrows);
int rvs_bfgs_result(void *h, double *x, double *fun, int32_t *nit, int32_t *nfev, int32_t *status, double *hess_inv, int64_t *rounds);
void rvs_bfgs_end(void *h);
}
int main() {
  const int S = 200, n = 6;
  std::vector<double> x0(S * n);
  for (int i = 0; i < S * n; i++) x0[i] = std::sin(0.37 * i) * 2;
  void *h = rvs_bfgs_begin(S, n, x0.data(), nullptr, 1e-5, 1e-4, 0.9, 0, 0);
  std::vector<int64_t> idx(S * (n + 1));
  std::vector<double> X(S * (n + 1) * n), F(S * (n + 1));
  int64_t rows;
  while ((rows = rvs_bfgs_pending(h, idx.data(), X.data(), S * (n + 1))) > 0) {
    for (int64_t r = 0; r < rows; r++) {
      double f = 0;
      for (int j = 0; j + 1 < n; j++) {
        const double a = X[r * n + j], b = X[r * n + j + 1];
        f += 100 * (b - a * a) * (b - a * a) + (1 - a) * (1 - a);
      }
      F[r] = f + ((idx[r] % 7 == 0) ? 1e-9 * std::sin(1e9 * X[r * n]) : 0.0);
    }
    if (rvs_bfgs_feed(h, F.data(), rows)) return 2;
  }
  std::vector<double> x(S * n), fun(S), H(S * n * n);
  std::vector<int32_t> nit(S), nfev(S), st(S);
  int64_t rounds;
  if (rvs_bfgs_result(h, x.data(), fun.data(), nit.data(), nfev.data(), st.data(), H.data(), &rounds)) return 3;
  rvs_bfgs_end(h);
  double fm = 0; int nmax = 0;
  for (int s = 0; s < S; s++) { fm += fun[s]; if (nit[s] > nmax) nmax = nit[s]; }
  printf("rounds %lld mean f %.3g max nit %d\n", (long long)rounds, fm / S, nmax);
  return 0;
}
