/*
 * oracle_core.c -- CPU restatement (plain C99) of the native / inner-loop parts
 * of the rvspecfit likelihood hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in rvspecfit_amd/ may link or call this:
 * it is the checker for the HIP kernels (tests/, __graft_entry__.smoke()) and
 * the "port" CPU baseline timed by bench.py.  Parity of this file against the
 * reference is pinned by tests/test_oracle_golden.py on vectors captured from
 * the reference itself (tests/golden/cases.npz).
 *
 * Reference rows restated here (paths relative to /root/reference):
 *   orc_spline_construct  py/rvspecfit/src/spliner.c:7-60    (natural cubic
 *                         spline, Thomas solve of the tridiagonal system)
 *   orc_spline_eval       py/rvspecfit/src/spliner.c:71-108  (O(1) knot index
 *                         on log- or lin-uniform knots + cubic evaluation)
 *   orc_chisq0            py/rvspecfit/spec_fit.py:203-303   (continuum
 *                         marginalisation: normal equations, Cholesky with an
 *                         eigen fallback standing in for the SVD branch)
 *   orc_chisq_vel         py/rvspecfit/spec_fit.py:707-727, 941-945 (Doppler
 *                         shift, spline evaluation and chisq0 for a list of
 *                         velocities of one arm)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_C_KMS 299792.458

/* spliner.c:7-60.  xs, ys: N knots; outputs have N-1 entries. */
void orc_spline_construct(const double *xs, const double *ys, int N, double *A,
                          double *B, double *C, double *D, double *h) {
  const int n1 = N - 1; /* intervals */
  double *z = (double *)calloc((size_t)N, sizeof(double));      /* 2nd derivs */
  double *slope = (double *)malloc((size_t)n1 * sizeof(double));
  double *hinv = (double *)malloc((size_t)n1 * sizeof(double));
  for (int i = 0; i < n1; i++) {
    h[i] = xs[i + 1] - xs[i];
    hinv[i] = 1. / h[i];
    slope[i] = (ys[i + 1] - ys[i]) * hinv[i];
  }
  /* interior system, unknowns z[1..N-2]:
   *   h[i-1] z[i-1] + 2 (h[i-1]+h[i]) z[i] + h[i] z[i+1] = 6 (slope[i]-slope[i-1])
   * Thomas forward sweep keeps the modified super-diagonal cp[] and rhs dp[]. */
  const int m = N - 2;
  if (m > 0) {
    double *cp = (double *)malloc((size_t)m * sizeof(double));
    double *dp = (double *)malloc((size_t)m * sizeof(double));
    double diag = 2 * (h[1] + h[0]);
    double rhs = 6 * (slope[1] - slope[0]);
    cp[0] = (m > 1) ? h[1] / diag : 0.;
    dp[0] = rhs / diag;
    for (int i = 1; i < m; i++) {
      diag = 2 * (h[i + 1] + h[i]);
      rhs = 6 * (slope[i + 1] - slope[i]);
      const double sub = h[i]; /* symmetric: sub-diagonal = previous super */
      const double den = diag - sub * cp[i - 1];
      if (i < m - 1) cp[i] = h[i + 1] / den;
      dp[i] = (rhs - sub * dp[i - 1]) / den;
    }
    z[m] = dp[m - 1];
    for (int i = m - 1; i >= 1; i--) z[i] = dp[i - 1] - cp[i - 1] * z[i + 1];
    free(cp);
    free(dp);
  }
  for (int i = 0; i < n1; i++) {
    const double a = hinv[i] * (1. / 6), b = h[i] * (1. / 6);
    A[i] = z[i + 1] * a;
    B[i] = z[i] * a;
    C[i] = ys[i + 1] * hinv[i] - z[i + 1] * b;
    D[i] = ys[i] * hinv[i] - z[i] * b;
  }
  free(z);
  free(slope);
  free(hinv);
}

/* spliner.c:71-108.  Returns 0, -1 (first/last eval point outside
 * [xs[0], xs[N-1])) or -2 (knots not uniform to 1e-10).  pos_out (nullable)
 * receives the integer interval index of every point. */
int orc_spline_eval(const double *evalx, int n, int N, const double *xs,
                    const double *hs, const double *A, const double *B,
                    const double *C, const double *D, int log_step,
                    double *ret, int *pos_out) {
  (void)hs;
  const double x0 = xs[0], xl = xs[N - 1];
  if (evalx[0] < x0 || evalx[n - 1] < x0) return -1;
  if (evalx[0] >= xl || evalx[n - 1] >= xl) return -1;
  double st, off;
  if (log_step) {
    st = log(xs[1] / x0);
    if (fabs(st - log(xs[2] / xs[1])) > 1e-10) return -2;
    off = log(x0);
  } else {
    st = xs[1] - x0;
    if (fabs(st - (xs[2] - xs[1])) > 1e-10) return -2;
    off = x0;
  }
  for (int i = 0; i < n; i++) {
    const double x = evalx[i];
    const int p = (int)(((log_step ? log(x) : x) - off) / st);
    const double dl = x - xs[p], dr = xs[p + 1] - x;
    ret[i] = A[p] * dl * dl * dl + B[p] * dr * dr * dr + C[p] * dl + D[p] * dr;
    if (pos_out) pos_out[i] = p;
  }
  return 0;
}

/* cyclic Jacobi eigenvalues/vectors of a small symmetric matrix (p<=32);
 * stands in for scipy.linalg.svd of the SPD normal matrix (spec_fit.py:288). */
static void jacobi_eig(double *M, int p, double *w, double *V) {
  for (int i = 0; i < p; i++)
    for (int j = 0; j < p; j++) V[i * p + j] = (i == j);
  for (int sweep = 0; sweep < 60; sweep++) {
    double offn = 0;
    for (int i = 0; i < p; i++)
      for (int j = i + 1; j < p; j++) offn += M[i * p + j] * M[i * p + j];
    if (offn < 1e-300) break;
    for (int a = 0; a < p; a++)
      for (int b = a + 1; b < p; b++) {
        const double apq = M[a * p + b];
        if (apq == 0) continue;
        const double th = (M[b * p + b] - M[a * p + a]) / (2 * apq);
        const double t = (th >= 0 ? 1. : -1.) / (fabs(th) + sqrt(th * th + 1));
        const double c = 1 / sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < p; k++) {
          const double ka = M[k * p + a], kb = M[k * p + b];
          M[k * p + a] = c * ka - s * kb;
          M[k * p + b] = s * ka + c * kb;
        }
        for (int k = 0; k < p; k++) {
          const double ak = M[a * p + k], bk = M[b * p + k];
          M[a * p + k] = c * ak - s * bk;
          M[b * p + k] = s * ak + c * bk;
        }
        for (int k = 0; k < p; k++) {
          const double ka = V[k * p + a], kb = V[k * p + b];
          V[k * p + a] = c * ka - s * kb;
          V[k * p + b] = s * ka + c * kb;
        }
      }
  }
  for (int i = 0; i < p; i++) w[i] = M[i * p + i];
}

/* spec_fit.py:203-303: -2 log L = log det Minv + 2 sum log e + |D - a^T ST|^2.
 * polys is [npoly][npix] row-major.  coeffs (nullable) receives a[npoly].
 * status (nullable): 0 Cholesky, 1 eigen fallback used. */
double orc_chisq0(const double *spec, const double *templ, const double *polys,
                  const double *espec, int npoly, int npix, double *coeffs,
                  int *status) {
  const int p = npoly;
  double *Dv = (double *)malloc((size_t)npix * sizeof(double));
  double *nt = (double *)malloc((size_t)npix * sizeof(double));
  double M[32 * 32], L[32 * 32], v[32], y[32], a[32];
  double logz = 0;
  for (int k = 0; k < npix; k++) {
    Dv[k] = spec[k] / espec[k];
    nt[k] = templ[k] / espec[k];
    logz += log(espec[k]);
  }
  for (int i = 0; i < p; i++) {
    double s = 0;
    for (int k = 0; k < npix; k++) s += polys[i * npix + k] * nt[k] * Dv[k];
    v[i] = s;
    for (int j = 0; j <= i; j++) {
      double g = 0;
      for (int k = 0; k < npix; k++)
        g += (polys[i * npix + k] * nt[k]) * (polys[j * npix + k] * nt[k]);
      M[i * p + j] = M[j * p + i] = g;
    }
  }
  int ok = 1;
  double ldet = 0;
  memset(L, 0, sizeof(L));
  for (int i = 0; i < p && ok; i++)
    for (int j = 0; j <= i; j++) {
      double s = M[i * p + j];
      for (int k = 0; k < j; k++) s -= L[i * p + k] * L[j * p + k];
      if (i == j) {
        if (!(s > 0)) {
          ok = 0;
          break;
        }
        L[i * p + i] = sqrt(s);
        ldet += 2 * log(L[i * p + i]);
      } else
        L[i * p + j] = s / L[j * p + j];
    }
  if (ok) {
    for (int i = 0; i < p; i++) {
      double s = v[i];
      for (int k = 0; k < i; k++) s -= L[i * p + k] * y[k];
      y[i] = s / L[i * p + i];
    }
    for (int i = p - 1; i >= 0; i--) {
      double s = y[i];
      for (int k = i + 1; k < p; k++) s -= L[k * p + i] * a[k];
      a[i] = s / L[i * p + i];
    }
  }
  double res = 0;
  if (ok) {
    for (int k = 0; k < npix; k++) {
      double m = 0;
      for (int i = 0; i < p; i++) m += a[i] * polys[i * npix + k];
      const double r = Dv[k] - m * nt[k];
      res += r * r;
    }
    if (!isfinite(ldet + res)) ok = 0;
  }
  if (!ok) {
    /* eigen-decomposition path == SVD of a symmetric matrix up to signs */
    double W[32 * 32], Vv[32 * 32], w[32];
    memcpy(W, M, sizeof(double) * 32 * 32);
    jacobi_eig(W, p, w, Vv);
    ldet = 0;
    for (int i = 0; i < p; i++) ldet += log(fabs(w[i]));
    for (int i = 0; i < p; i++) {
      double s = 0;
      for (int j = 0; j < p; j++) {
        double vj = 0;
        for (int k = 0; k < p; k++) vj += Vv[k * p + j] * v[k];
        s += Vv[i * p + j] * vj / w[j];
      }
      a[i] = s;
    }
    res = 0;
    for (int k = 0; k < npix; k++) {
      double m = 0;
      for (int i = 0; i < p; i++) m += a[i] * polys[i * npix + k];
      const double r = Dv[k] - m * nt[k];
      res += r * r;
    }
  }
  if (coeffs) memcpy(coeffs, a, sizeof(double) * (size_t)p);
  if (status) *status = ok ? 0 : 1;
  free(Dv);
  free(nt);
  return ldet + 2 * logz + res;
}

/* spec_fit.py:707-727 + 912 + 941: for every velocity, Doppler-shift the
 * observed wavelengths, evaluate the template spline there and marginalise the
 * continuum.  Returns the first non-zero spline status (0 if none). */
int orc_chisq_vel(const double *lam, const double *spec, const double *espec,
                  const double *polys, int npoly, int npix, const double *xs,
                  const double *hs, const double *A, const double *B,
                  const double *C, const double *D, int N, int log_step,
                  const double *vels, int nvel, double *out) {
  double *ex = (double *)malloc((size_t)npix * sizeof(double));
  double *tm = (double *)malloc((size_t)npix * sizeof(double));
  int rc = 0;
  for (int iv = 0; iv < nvel; iv++) {
    const double beta = vels[iv] / ORC_C_KMS;
    const double f = sqrt((1 - beta) / (1 + beta));
    for (int k = 0; k < npix; k++) ex[k] = lam[k] * f;
    const int st =
        orc_spline_eval(ex, npix, N, xs, hs, A, B, C, D, log_step, tm, 0);
    if (st != 0) {
      out[iv] = NAN;
      if (!rc) rc = st;
      continue;
    }
    out[iv] = orc_chisq0(spec, tm, polys, espec, npoly, npix, 0, 0);
  }
  free(ex);
  free(tm);
  return rc;
}
