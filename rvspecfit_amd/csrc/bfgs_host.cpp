// bfgs_host.cpp -- host side of the second minimiser of vel_fit.process
// (vel_fit.py:653-658: scipy.optimize.minimize(method='BFGS', hess_inv0=...)).
//
// S independent BFGS runs advance in lock-step: every run is a C++20 coroutine
// that suspends whenever it needs objective values (one point, the n forward-
// difference points of a gradient, or both), the driver gathers the requests of
// all runs into ONE batch for the GPU objective and resumes them with the
// values.  The algorithm is scipy's, statement for statement:
//   _minimize_bfgs (scipy/optimize/_optimize.py), ScalarFunction's caching of
//   f/g at the latest x, approx_derivative(method='2-point', abs_step=1.49e-8),
//   line_search_wolfe1 = MINPACK-2 dcsrch/dcstep (_dcsrch.py), and the
//   line_search_wolfe2/_zoom fall-back (_linesearch.py).
// rvspecfit_amd/bfgs.py is the same restatement in Python and is the one pinned
// bit for bit against scipy (tests/test_tools_cpu.py); this file repeats it in
// scalar C++ (products and sums in index order, no FMA contraction), which
// agrees with numpy's BLAS-backed dot products to rounding, not to the bit.
// It exists because the Python generators cost ~20 us per resume: 2.6 s per
// 2000 spectra, more than the Nelder-Mead stage on the GPU.
#include <cmath>
#include <coroutine>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/rvsgpu.h"

namespace {

constexpr int MAXN = 16;
constexpr double EPS_FD = 1.4901161193847656e-08;  // sqrt(DBL_EPSILON)

struct Task {
  struct promise_type {
    Task get_return_object() {
      return Task{std::coroutine_handle<promise_type>::from_promise(*this)};
    }
    std::suspend_always initial_suspend() noexcept { return {}; }
    std::suspend_always final_suspend() noexcept { return {}; }
    void return_void() {}
    void unhandled_exception() {}
  };
  std::coroutine_handle<promise_type> h;
};

struct Run {
  int n = 0;
  // ---- request / reply mailbox
  int nrows = 0;                   // 0: nothing pending
  double rows[(MAXN + 1) * MAXN];  // points to evaluate
  double vals[MAXN + 1];           // their values
  // ---- ScalarFunction cache
  bool has_x = false, has_f = false, has_g = false;
  double sx[MAXN], f = 0, g[MAXN];
  int nfev = 0, ngev = 0;
  // ---- parameters
  double gtol, c1, c2, xrtol;
  int maxiter;
  double x0[MAXN], H0[MAXN * MAXN];
  // ---- result
  double xk[MAXN], fval = 0, Hk[MAXN * MAXN], gfk[MAXN];
  int nit = 0, status = 0;
  Task task{};
  bool done = false;
};

inline double dot(const double *a, const double *b, int n) {
  double s = 0;
  for (int i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}
inline double sgn(double x) { return x == x ? (double)((x > 0) - (x < 0)) : x; }
inline double max3(double a, double b, double c) {
  // Python's max(): first maximal element, nan-insensitive comparisons
  double m = a;
  if (b > m) m = b;
  if (c > m) m = c;
  return m;
}

// ---- ScalarFunction -------------------------------------------------------
void sf_set_x(Run &c, const double *x) {
  bool same = c.has_x;
  if (same)
    for (int i = 0; i < c.n; i++)
      if (!(x[i] == c.sx[i])) same = false;
  if (!same) {
    std::memcpy(c.sx, x, sizeof(double) * c.n);
    c.has_x = true;
    c.has_f = c.has_g = false;
  }
}
void fd_points(const Run &c, double *out) {  // [n, n]
  const int n = c.n;
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) out[i * n + j] = c.sx[j];
    const double dx = (c.sx[i] + EPS_FD) - c.sx[i];
    double h = EPS_FD;
    bool anyzero = false;
    for (int j = 0; j < n; j++)
      if ((c.sx[j] + EPS_FD) - c.sx[j] == 0) anyzero = true;
    if (anyzero && dx == 0)
      h = EPS_FD * (c.sx[i] >= 0 ? 1.0 : -1.0) *
          std::fmax(1.0, std::fabs(c.sx[i]));
    out[i * n + i] = c.sx[i] + h;
  }
}
void req_f(Run &c) {
  std::memcpy(c.rows, c.sx, sizeof(double) * c.n);
  c.nrows = 1;
}
void req_g(Run &c) {
  fd_points(c, c.rows);
  c.nrows = c.n;
}
void fin_g(Run &c, const double *f1, const double *x1) {
  for (int i = 0; i < c.n; i++)
    c.g[i] = (f1[i] - c.f) / (x1[i * c.n + i] - c.sx[i]);
  c.has_g = true;
  c.nfev += c.n;
  c.ngev += 1;
}
void req_fg(Run &c) {
  std::memcpy(c.rows, c.sx, sizeof(double) * c.n);
  fd_points(c, c.rows + c.n);
  c.nrows = c.n + 1;
}

#define SUSPEND() co_await std::suspend_always {}
#define SF_FUN(xv, out)         \
  do {                          \
    sf_set_x(c, xv);            \
    if (!c.has_f) {             \
      req_f(c);                 \
      SUSPEND();                \
      c.f = c.vals[0];          \
      c.has_f = true;           \
      c.nfev += 1;              \
    }                           \
    out = c.f;                  \
  } while (0)
#define SF_GRAD(xv)                 \
  do {                              \
    sf_set_x(c, xv);                \
    if (!c.has_g) {                 \
      if (!c.has_f) {               \
        req_f(c);                   \
        SUSPEND();                  \
        c.f = c.vals[0];            \
        c.has_f = true;             \
        c.nfev += 1;                \
      }                             \
      req_g(c);                     \
      SUSPEND();                    \
      fin_g(c, c.vals, c.rows);     \
    }                               \
  } while (0)
#define SF_FUN_GRAD(xv, out)              \
  do {                                    \
    sf_set_x(c, xv);                      \
    if (!c.has_f && !c.has_g) {           \
      req_fg(c);                          \
      SUSPEND();                          \
      c.f = c.vals[0];                    \
      c.has_f = true;                     \
      c.nfev += 1;                        \
      fin_g(c, c.vals + 1, c.rows + c.n); \
      out = c.f;                          \
    } else {                              \
      SF_FUN(xv, out);                    \
      SF_GRAD(xv);                        \
    }                                     \
  } while (0)

// ---- MINPACK-2 dcstep (scipy/optimize/_dcsrch.py) ------------------------------
struct StepState {
  double stx, fx, dx, sty, fy, dy, stp;
  bool brackt;
};
void dcstep(StepState &s, double fp, double dp, double stpmin, double stpmax) {
  double stx = s.stx, fx = s.fx, dx = s.dx, sty = s.sty, fy = s.fy, dy = s.dy,
         stp = s.stp;
  bool brackt = s.brackt;
  const double sgnd = sgn(dp) * sgn(dx);
  double stpf;
  if (fp > fx) {
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double ss = max3(std::fabs(theta), std::fabs(dx), std::fabs(dp));
    double gamma =
        ss * std::sqrt((theta / ss) * (theta / ss) - (dx / ss) * (dp / ss));
    if (stp < stx) gamma *= -1;
    const double p = (gamma - dx) + theta;
    const double q = ((gamma - dx) + gamma) + dp;
    const double r = p / q;
    const double stpc = stx + r * (stp - stx);
    const double stpq =
        stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
    if (std::fabs(stpc - stx) <= std::fabs(stpq - stx))
      stpf = stpc;
    else
      stpf = stpc + (stpq - stpc) / 2.0;
    brackt = true;
  } else if (sgnd < 0.0) {
    const double theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
    const double ss = max3(std::fabs(theta), std::fabs(dx), std::fabs(dp));
    double gamma =
        ss * std::sqrt((theta / ss) * (theta / ss) - (dx / ss) * (dp / ss));
    if (stp > stx) gamma *= -1;
    const double p = (gamma - dp) + theta;
    const double q = ((gamma - dp) + gamma) + dx;
    const double r = p / q;
    const double stpc = stp + r * (stx - stp);
    const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (std::fabs(stpc - stp) > std::fabs(stpq - stp))
      stpf = stpc;
    else
      stpf = stpq;
    brackt = true;
  } else if (std::fabs(dp) < std::fabs(dx)) {
    const double theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
    const double ss = max3(std::fabs(theta), std::fabs(dx), std::fabs(dp));
    const double rad = (theta / ss) * (theta / ss) - (dx / ss) * (dp / ss);
    double gamma = ss * std::sqrt((rad > 0) ? rad : 0.0);  // max(0, rad)
    if (stp > stx) gamma = -gamma;
    const double p = (gamma - dp) + theta;
    const double q = (gamma + (dx - dp)) + gamma;
    const double r = p / q;
    double stpc;
    if (r < 0 && gamma != 0)
      stpc = stp + r * (stx - stp);
    else if (stp > stx)
      stpc = stpmax;
    else
      stpc = stpmin;
    const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (brackt) {
      if (std::fabs(stpc - stp) < std::fabs(stpq - stp))
        stpf = stpc;
      else
        stpf = stpq;
      const double lim = stp + 0.66 * (sty - stp);
      if (stp > stx)
        stpf = (stpf < lim) ? stpf : lim;  // min(lim, stpf)
      else
        stpf = (stpf > lim) ? stpf : lim;  // max(lim, stpf)
    } else {
      if (std::fabs(stpc - stp) > std::fabs(stpq - stp))
        stpf = stpc;
      else
        stpf = stpq;
      // min(max(stpf, stpmin), stpmax) with Python's comparison semantics
      double t = (stpmin > stpf) ? stpmin : stpf;
      stpf = (stpmax < t) ? stpmax : t;
    }
  } else {
    if (brackt) {
      const double theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
      const double ss = max3(std::fabs(theta), std::fabs(dy), std::fabs(dp));
      double gamma =
          ss * std::sqrt((theta / ss) * (theta / ss) - (dy / ss) * (dp / ss));
      if (stp > sty) gamma = -gamma;
      const double p = (gamma - dp) + theta;
      const double q = ((gamma - dp) + gamma) + dy;
      const double r = p / q;
      stpf = stp + r * (sty - stp);
    } else if (stp > stx) {
      stpf = stpmax;
    } else {
      stpf = stpmin;
    }
  }
  if (fp > fx) {
    sty = stp;
    fy = fp;
    dy = dp;
  } else {
    if (sgnd < 0) {
      sty = stx;
      fy = fx;
      dy = dx;
    }
    stx = stp;
    fx = fp;
    dx = dp;
  }
  s.stx = stx, s.fx = fx, s.dx = dx, s.sty = sty, s.fy = fy, s.dy = dy;
  s.stp = stpf;
  s.brackt = brackt;
}

enum { T_FG = 0, T_ERROR, T_WARN, T_CONV };

struct Dcsrch {
  double ftol, gtol, xtol, stpmin, stpmax;
  bool started = false, brackt = false;
  int stage = 1;
  double finit, ginit, gtest, width, width1, stx, fx, gx, sty, fy, gy, stmin,
      stmax;
  // returns task; stp updated in place
  int step(double &stp, double f, double g) {
    const double p5 = 0.5, p66 = 0.66, xtrapl = 1.1, xtrapu = 4.0;
    if (!started) {
      started = true;
      int task = T_FG;
      if (stp < stpmin) task = T_ERROR;
      if (stp > stpmax) task = T_ERROR;
      if (g >= 0) task = T_ERROR;
      if (task == T_ERROR) return task;
      brackt = false;
      stage = 1;
      finit = f, ginit = g;
      gtest = ftol * ginit;
      width = stpmax - stpmin;
      width1 = width / p5;
      stx = 0.0, fx = finit, gx = ginit;
      sty = 0.0, fy = finit, gy = ginit;
      stmin = 0;
      stmax = stp + xtrapu * stp;
      return T_FG;
    }
    int task = T_FG;
    const double ftest = finit + stp * gtest;
    if (stage == 1 && f <= ftest && g >= 0) stage = 2;
    if (brackt && (stp <= stmin || stp >= stmax)) task = T_WARN;
    if (brackt && stmax - stmin <= xtol * stmax) task = T_WARN;
    if (stp == stpmax && f <= ftest && g <= gtest) task = T_WARN;
    if (stp == stpmin && (f > ftest || g >= gtest)) task = T_WARN;
    if (f <= ftest && std::fabs(g) <= gtol * -ginit) task = T_CONV;
    if (task != T_FG) return task;
    StepState s;
    if (stage == 1 && f <= fx && f > ftest) {
      const double fm = f - stp * gtest;
      double fxm = fx - stx * gtest, fym = fy - sty * gtest;
      const double gm = g - gtest;
      double gxm = gx - gtest, gym = gy - gtest;
      s = {stx, fxm, gxm, sty, fym, gym, stp, brackt};
      dcstep(s, fm, gm, stmin, stmax);
      stx = s.stx, sty = s.sty, stp = s.stp, brackt = s.brackt;
      fxm = s.fx, gxm = s.dx, fym = s.fy, gym = s.dy;
      fx = fxm + stx * gtest;
      fy = fym + sty * gtest;
      gx = gxm + gtest;
      gy = gym + gtest;
    } else {
      s = {stx, fx, gx, sty, fy, gy, stp, brackt};
      dcstep(s, f, g, stmin, stmax);
      stx = s.stx, fx = s.fx, gx = s.dx, sty = s.sty, fy = s.fy, gy = s.dy;
      stp = s.stp, brackt = s.brackt;
    }
    if (brackt) {
      if (std::fabs(sty - stx) >= p66 * width1) stp = stx + p5 * (sty - stx);
      width1 = width;
      width = std::fabs(sty - stx);
    }
    if (brackt) {
      stmin = (sty < stx) ? sty : stx;  // min(stx, sty)
      stmax = (sty > stx) ? sty : stx;  // max(stx, sty)
    } else {
      stmin = stp + xtrapl * (stp - stx);
      stmax = stp + xtrapu * (stp - stx);
    }
    {
      double t = (stpmin > stp) ? stpmin : stp;  // _clip
      stp = (stpmax < t) ? stpmax : t;
    }
    if ((brackt && (stp <= stmin || stp >= stmax)) ||
        (brackt && stmax - stmin <= xtol * stmax))
      stp = stx;
    return T_FG;
  }
};

// ---- _cubicmin / _quadmin (scipy/optimize/_linesearch.py); false = None --------
bool cubicmin(double a, double fa, double fpa, double b, double fb, double c,
              double fc, double &xmin) {
  const double C = fpa;
  const double db = b - a, dc = c - a;
  const double denom = (db * dc) * (db * dc) * (db - dc);
  if (denom == 0) return false;  // the division raises under errstate('raise')
  // Python floats: x**3 is C pow()
  const double d00 = dc * dc, d01 = -(db * db), d10 = -std::pow(dc, 3.0),
               d11 = std::pow(db, 3.0);
  const double v0 = fb - fa - C * db, v1 = fc - fa - C * dc;
  double A = d00 * v0 + d01 * v1, B = d10 * v0 + d11 * v1;
  A /= denom;
  B /= denom;
  const double radical = B * B - 3 * A * C;
  if (!(radical >= 0)) return false;  // sqrt: invalid
  if (3 * A == 0) return false;       // division by zero
  xmin = a + (-B + std::sqrt(radical)) / (3 * A);
  return std::isfinite(xmin) && std::isfinite(A) && std::isfinite(B);
}
bool quadmin(double a, double fa, double fpa, double b, double fb,
             double &xmin) {
  const double D = fa, C = fpa, db = b - a * 1.0;
  if (db * db == 0) return false;
  const double B = (fb - D - C * db) / (db * db);
  if (2.0 * B == 0) return false;
  xmin = a - C / (2.0 * B);
  return std::isfinite(xmin) && std::isfinite(B);
}

// ---- one BFGS run -------------------------------------------------------------
Task bfgs_run(Run &c) {
  const int n = c.n;
  double xk[MAXN], gfk[MAXN], pk[MAXN], xt[MAXN], gfkp1[MAXN], sk[MAXN],
      yk[MAXN];
  double Hk[MAXN * MAXN];
  std::memcpy(xk, c.x0, sizeof(double) * n);
  std::memcpy(Hk, c.H0, sizeof(double) * n * n);
  double old_fval;
  SF_FUN_GRAD(xk, old_fval);
  std::memcpy(gfk, c.g, sizeof(double) * n);
  int k = 0, warnflag = 0;
  // np.linalg.norm: sqrt of the sum of squares
  double old_old_fval = old_fval + std::sqrt(dot(gfk, gfk, n)) / 2;
  bool have_old_old = true;
  double gnorm = 0;
  for (int i = 0; i < n; i++) {
    const double a = std::fabs(gfk[i]);
    if (a > gnorm || a != a) gnorm = a;  // np.amax propagates nan
  }
  while (gnorm > c.gtol && k < c.maxiter) {
    for (int i = 0; i < n; i++) pk[i] = -dot(Hk + i * n, gfk, n);
    double alpha_k = 0, fval = 0, ofv = 0;
    bool have_stp = false, have_gnew = false;
    // ---------------- line_search_wolfe1 (amin=1e-100, amax=1e100) ----------
    {
      const double derphi0 = dot(gfk, pk, n);
      const double phi0 = old_fval;
      double alpha1;
      if (have_old_old && derphi0 != 0) {
        alpha1 = 1.01 * 2 * (phi0 - old_old_fval) / derphi0;
        alpha1 = (alpha1 < 1.0) ? alpha1 : 1.0;  // min(1.0, alpha1)
        if (alpha1 < 0) alpha1 = 1.0;
      } else {
        alpha1 = 1.0;
      }
      Dcsrch ds{c.c1, c.c2, 1e-14, 1e-100, 1e100};
      double phi1 = phi0, derphi1 = derphi0, stp = alpha1;
      std::memcpy(gfkp1, gfk, sizeof(double) * n);
      int task = T_FG;
      bool stp_ok = false;
      int it = 0;
      for (; it < 100; it++) {
        stp = alpha1;
        task = ds.step(stp, phi1, derphi1);
        if (!std::isfinite(stp)) {
          task = T_WARN;
          stp_ok = false;
          break;
        }
        stp_ok = true;
        if (task == T_FG) {
          alpha1 = stp;
          for (int i = 0; i < n; i++) xt[i] = xk[i] + stp * pk[i];
          SF_FUN_GRAD(xt, phi1);
          std::memcpy(gfkp1, c.g, sizeof(double) * n);
          derphi1 = dot(gfkp1, pk, n);
        } else {
          break;
        }
      }
      if (it == 100) {
        stp_ok = false;
        task = T_WARN;
      }
      if (task == T_ERROR || task == T_WARN) stp_ok = false;
      if (stp_ok) {
        have_stp = true;
        alpha_k = stp;
        fval = phi1;
        ofv = phi0;
        have_gnew = true;
      }
    }
    // ---------------- line_search_wolfe2 fall-back ---------------------------
    if (!have_stp) {
      const double amax = 1e100, c1 = c.c1, c2 = c.c2;
      const double derphi0 = dot(gfk, pk, n);
      double phi0 = old_fval;
      const double old_phi0 = old_old_fval;
      double alpha0 = 0, alpha1;
      if (have_old_old && derphi0 != 0) {
        alpha1 = 1.01 * 2 * (phi0 - old_phi0) / derphi0;
        alpha1 = (alpha1 < 1.0) ? alpha1 : 1.0;
      } else {
        alpha1 = 1.0;
      }
      if (alpha1 < 0) alpha1 = 1.0;
      alpha1 = (amax < alpha1) ? amax : alpha1;
      double phi_a1;
      for (int i = 0; i < n; i++) xt[i] = xk[i] + alpha1 * pk[i];
      SF_FUN(xt, phi_a1);
      double phi_a0 = phi0, derphi_a0 = derphi0;
      bool star_alpha = false, star_der = false;
      double alpha_star = 0, phi_star = 0;
      // zoom arguments
      bool do_zoom = false;
      double z_lo = 0, z_hi = 0, zphi_lo = 0, zphi_hi = 0, zder_lo = 0;
      bool fell_through = true;
      for (int i = 0; i < 10; i++) {
        if (alpha1 == 0 || alpha0 > amax) {
          star_alpha = false;
          phi_star = phi0;
          star_der = false;
          phi0 = old_phi0;
          fell_through = false;
          break;
        }
        if ((phi_a1 > phi0 + c1 * alpha1 * derphi0) ||
            ((phi_a1 >= phi_a0) && i > 0)) {
          do_zoom = true;
          z_lo = alpha0, z_hi = alpha1, zphi_lo = phi_a0, zphi_hi = phi_a1,
          zder_lo = derphi_a0;
          fell_through = false;
          break;
        }
        for (int q = 0; q < n; q++) xt[q] = xk[q] + alpha1 * pk[q];
        SF_GRAD(xt);
        std::memcpy(gfkp1, c.g, sizeof(double) * n);
        const double derphi_a1 = dot(gfkp1, pk, n);
        if (std::fabs(derphi_a1) <= -c2 * derphi0) {
          star_alpha = true;
          alpha_star = alpha1;
          phi_star = phi_a1;
          star_der = true;
          fell_through = false;
          break;
        }
        if (derphi_a1 >= 0) {
          do_zoom = true;
          z_lo = alpha1, z_hi = alpha0, zphi_lo = phi_a1, zphi_hi = phi_a0,
          zder_lo = derphi_a1;
          fell_through = false;
          break;
        }
        double alpha2 = 2 * alpha1;
        alpha2 = (amax < alpha2) ? amax : alpha2;
        alpha0 = alpha1;
        alpha1 = alpha2;
        phi_a0 = phi_a1;
        for (int q = 0; q < n; q++) xt[q] = xk[q] + alpha1 * pk[q];
        SF_FUN(xt, phi_a1);
        derphi_a0 = derphi_a1;
      }
      if (fell_through) {  // the for-else of scalar_search_wolfe2
        star_alpha = true;
        alpha_star = alpha1;
        phi_star = phi_a1;
        star_der = false;
      }
      if (do_zoom) {
        double a_lo = z_lo, a_hi = z_hi, phi_lo = zphi_lo, phi_hi = zphi_hi,
               derphi_lo = zder_lo;
        int i = 0;
        const double delta1 = 0.2, delta2 = 0.1;
        double phi_rec = phi0, a_rec = 0;
        star_alpha = false;
        star_der = false;
        while (true) {
          const double dalpha = a_hi - a_lo;
          double a, b;
          if (dalpha < 0)
            a = a_hi, b = a_lo;
          else
            a = a_lo, b = a_hi;
          double a_j = 0, cchk = 0;
          bool have_aj = false;
          if (i > 0) {
            cchk = delta1 * dalpha;
            have_aj = cubicmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi, a_rec,
                               phi_rec, a_j);
          }
          if (i == 0 || !have_aj || a_j > b - cchk || a_j < a + cchk) {
            const double qchk = delta2 * dalpha;
            have_aj = quadmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi, a_j);
            if (!have_aj || a_j > b - qchk || a_j < a + qchk)
              a_j = a_lo + 0.5 * dalpha;
          }
          double phi_aj;
          for (int q = 0; q < n; q++) xt[q] = xk[q] + a_j * pk[q];
          SF_FUN(xt, phi_aj);
          if ((phi_aj > phi0 + c1 * a_j * derphi0) || (phi_aj >= phi_lo)) {
            phi_rec = phi_hi, a_rec = a_hi;
            a_hi = a_j, phi_hi = phi_aj;
          } else {
            for (int q = 0; q < n; q++) xt[q] = xk[q] + a_j * pk[q];
            SF_GRAD(xt);
            std::memcpy(gfkp1, c.g, sizeof(double) * n);
            const double derphi_aj = dot(gfkp1, pk, n);
            if (std::fabs(derphi_aj) <= -c2 * derphi0) {
              star_alpha = true;
              alpha_star = a_j;
              phi_star = phi_aj;
              star_der = true;
              break;
            }
            if (derphi_aj * (a_hi - a_lo) >= 0) {
              phi_rec = phi_hi, a_rec = a_hi;
              a_hi = a_lo, phi_hi = phi_lo;
            } else {
              phi_rec = phi_lo, a_rec = a_lo;
            }
            a_lo = a_j, phi_lo = phi_aj, derphi_lo = derphi_aj;
          }
          i += 1;
          if (i > 10) break;  // (None, None, None)
        }
      }
      if (star_alpha) {
        have_stp = true;
        alpha_k = alpha_star;
        fval = phi_star;
        ofv = phi0;
        have_gnew = star_der;  // gval[0] of the last derphi call
      }
    }
    if (!have_stp) {
      warnflag = 2;
      break;
    }
    old_fval = fval;
    old_old_fval = ofv;
    have_old_old = true;
    for (int i = 0; i < n; i++) {
      sk[i] = alpha_k * pk[i];
      xk[i] = xk[i] + sk[i];
    }
    if (!have_gnew) {
      SF_GRAD(xk);
      std::memcpy(gfkp1, c.g, sizeof(double) * n);
    }
    for (int i = 0; i < n; i++) {
      yk[i] = gfkp1[i] - gfk[i];
      gfk[i] = gfkp1[i];
    }
    k += 1;
    gnorm = 0;
    for (int i = 0; i < n; i++) {
      const double a = std::fabs(gfk[i]);
      if (a > gnorm || a != a) gnorm = a;
    }
    if (gnorm <= c.gtol) break;
    {
      double pp = 0, xx = 0;
      for (int i = 0; i < n; i++) pp += pk[i] * pk[i];
      for (int i = 0; i < n; i++) xx += xk[i] * xk[i];
      if (alpha_k * std::sqrt(pp) <= c.xrtol * (c.xrtol + std::sqrt(xx))) break;
    }
    if (!std::isfinite(old_fval)) {
      warnflag = 2;
      break;
    }
    const double rhok_inv = dot(yk, sk, n);
    const double rhok = (rhok_inv == 0.) ? 1000.0 : 1. / rhok_inv;
    // Hk = A1 Hk A2 + rhok sk sk^T, A1 = I - sk yk^T rhok, A2 = I - yk sk^T rhok
    double A1[MAXN * MAXN], A2[MAXN * MAXN], T1[MAXN * MAXN];
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        A1[i * n + j] = (i == j ? 1.0 : 0.0) - sk[i] * yk[j] * rhok;
        A2[i * n + j] = (i == j ? 1.0 : 0.0) - yk[i] * sk[j] * rhok;
      }
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        double s = 0;
        for (int q = 0; q < n; q++) s += Hk[i * n + q] * A2[q * n + j];
        T1[i * n + j] = s;
      }
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        double s = 0;
        for (int q = 0; q < n; q++) s += A1[i * n + q] * T1[q * n + j];
        Hk[i * n + j] = s + (rhok * sk[i]) * sk[j];
      }
  }
  double fval = old_fval;
  if (warnflag == 2) {
  } else if (k >= c.maxiter) {
    warnflag = 1;
  } else {
    bool xnan = false;
    for (int i = 0; i < n; i++)
      if (xk[i] != xk[i]) xnan = true;
    if (gnorm != gnorm || fval != fval || xnan) warnflag = 3;
  }
  std::memcpy(c.xk, xk, sizeof(double) * n);
  std::memcpy(c.gfk, gfk, sizeof(double) * n);
  std::memcpy(c.Hk, Hk, sizeof(double) * n * n);
  c.fval = fval;
  c.nit = k;
  c.status = warnflag;
  c.nrows = 0;
  c.done = true;
  co_return;
}

struct Driver {
  int S, n;
  std::vector<Run> runs;
  std::vector<int> order;  // spectra with a pending request, ascending
  int64_t rounds = 0;
};

void advance(Run &r) {
  r.nrows = 0;
  r.task.h.resume();
}

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
    r.n = n;
    r.gtol = gtol, r.c1 = c1, r.c2 = c2, r.xrtol = xrtol;
    r.maxiter = maxiter > 0 ? maxiter : n * 200;
    std::memcpy(r.x0, x0 + (int64_t)s * n, sizeof(double) * n);
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++)
        r.H0[i * n + j] = hess_inv0 ? hess_inv0[i * n + j] : (i == j ? 1.0 : 0.0);
  }
  // the coroutine frames hold references into d->runs: no resize after this
  for (int s = 0; s < S; s++) {
    Run &r = d->runs[s];
    r.task = bfgs_run(r);
    advance(r);  // runs to the first request (f and g at x0)
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
    advance(r);
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
  for (Run &r : d->runs)
    if (r.task.h) r.task.h.destroy();
  delete d;
}

}  // extern "C"
