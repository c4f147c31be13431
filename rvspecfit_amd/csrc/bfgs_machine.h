// bfgs_machine.h -- ONE run of the second minimiser of vel_fit.process
// (vel_fit.py:653-658: scipy.optimize.minimize(method='BFGS', hess_inv0=...)) as a
// resumable state machine: `advance(run)` executes until the run needs objective
// values (one point, the n forward-difference points of a gradient, or both),
// leaves the points in run.rows / run.nrows and returns; the caller evaluates them,
// writes run.vals and calls advance again.  run.done ends it.
//
// The same source runs on the host (bfgs_host.cpp: rvs_bfgs_begin / _pending /
// _feed, what the CPU suite pins to the scipy-pinned Python restatement) and in a
// kernel, one thread per spectrum (bfgs_dev.hip: rvs_bfgs_run), so the device runs
// the pinned statement, not a copy of it.  The algorithm is scipy's, statement for
// statement:
//   _minimize_bfgs (scipy/optimize/_optimize.py), ScalarFunction's caching of
//   f/g at the latest x, approx_derivative(method='2-point', abs_step=1.49e-8),
//   line_search_wolfe1 = MINPACK-2 dcsrch/dcstep (_dcsrch.py), and the
//   line_search_wolfe2/_zoom fall-back (_linesearch.py).
// Scalar arithmetic in index order, no FMA contraction (a*b+c rounds twice, as
// numpy's element-wise arithmetic does): it follows numpy's BLAS-backed dot
// products to rounding, not to the bit.
//
// Form: a protothread.  Everything that lives across a suspension is a member of
// Run; BF_YIELD() records the resume point and returns, the switch at the top of
// advance() jumps back to it (into the loops: that is what the construct is for).
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define BF_HD __host__ __device__
#else
#define BF_HD
#endif
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace rvs_bfgs {

constexpr int MAXN = 16;
constexpr double EPS_FD = 1.4901161193847656e-08;  // sqrt(DBL_EPSILON)

enum { T_FG = 0, T_ERROR, T_WARN, T_CONV };

// ---- MINPACK-2 dcstep (scipy/optimize/_dcsrch.py) ------------------------------
struct StepState {
  double stx, fx, dx, sty, fy, dy, stp;
  bool brackt;
};

BF_HD inline double sgn(double x) {
  return x == x ? (double)((x > 0) - (x < 0)) : x;
}
BF_HD inline double max3(double a, double b, double c) {
  // Python's max(): first maximal element, nan-insensitive comparisons
  double m = a;
  if (b > m) m = b;
  if (c > m) m = c;
  return m;
}

BF_HD inline void dcstep(StepState &s, double fp, double dp, double stpmin,
                         double stpmax) {
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

// ---- MINPACK-2 dcsrch (scipy/optimize/_dcsrch.py: DCSRCH.__call__'s body) --------
struct Dcsrch {
  double ftol, gtol, xtol, stpmin, stpmax;
  bool started, brackt;
  int stage;
  double finit, ginit, gtest, width, width1, stx, fx, gx, sty, fy, gy, stmin,
      stmax;
  BF_HD void reset(double ftol_, double gtol_, double xtol_, double stpmin_,
                   double stpmax_) {
    ftol = ftol_, gtol = gtol_, xtol = xtol_, stpmin = stpmin_, stpmax = stpmax_;
    started = false, brackt = false, stage = 1;
  }
  // returns task; stp updated in place
  BF_HD int step(double &stp, double f, double g) {
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
      s.stx = stx, s.fx = fxm, s.dx = gxm, s.sty = sty, s.fy = fym, s.dy = gym;
      s.stp = stp, s.brackt = brackt;
      dcstep(s, fm, gm, stmin, stmax);
      stx = s.stx, sty = s.sty, stp = s.stp, brackt = s.brackt;
      fxm = s.fx, gxm = s.dx, fym = s.fy, gym = s.dy;
      fx = fxm + stx * gtest;
      fy = fym + sty * gtest;
      gx = gxm + gtest;
      gy = gym + gtest;
    } else {
      s.stx = stx, s.fx = fx, s.dx = gx, s.sty = sty, s.fy = fy, s.dy = gy;
      s.stp = stp, s.brackt = brackt;
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
BF_HD inline bool cubicmin(double a, double fa, double fpa, double b, double fb,
                           double c, double fc, double &xmin) {
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
BF_HD inline bool quadmin(double a, double fa, double fpa, double b, double fb,
                          double &xmin) {
  const double D = fa, C = fpa, db = b - a * 1.0;
  if (db * db == 0) return false;
  const double B = (fb - D - C * db) / (db * db);
  if (2.0 * B == 0) return false;
  xmin = a - C / (2.0 * B);
  return std::isfinite(xmin) && std::isfinite(B);
}

// ---- one run: its mailbox, ScalarFunction's cache, and every local of scipy's
// functions that lives across an objective call ----------------------------------
struct Run {
  int n;
  int pc;  // resume point of advance() (0: not started)
  // ---- request / reply mailbox
  int nrows;                       // 0: nothing pending
  double rows[(MAXN + 1) * MAXN];  // points to evaluate
  double vals[MAXN + 1];           // their values
  // ---- ScalarFunction cache
  bool has_x, has_f, has_g;
  double sx[MAXN], f, g[MAXN];
  int nfev, ngev;
  // ---- parameters
  double gtol, c1, c2, xrtol;
  int maxiter;
  // ---- state of _minimize_bfgs (xk / Hk start as x0 / hess_inv0)
  double xk[MAXN], gfk[MAXN], pk[MAXN], xt[MAXN], gfkp1[MAXN], sk[MAXN], yk[MAXN];
  double Hk[MAXN * MAXN], T1[MAXN * MAXN];
  int k, warnflag;
  double old_fval, old_old_fval, gnorm, alpha_k, fval, ofv;
  bool have_old_old, have_stp, have_gnew;
  // ---- line_search_wolfe1
  Dcsrch ds;
  double derphi0, phi0, alpha1, phi1, derphi1, stp;
  int task, it;
  bool stp_ok;
  // ---- line_search_wolfe2 / _zoom
  double old_phi0, alpha0, phi_a1, phi_a0, derphi_a0, derphi_a1, alpha_star, phi_star;
  bool star_alpha, star_der, do_zoom, fell_through;
  double z_lo, z_hi, zphi_lo, zphi_hi, zder_lo;
  int i2;
  double a_lo, a_hi, phi_lo, phi_hi, derphi_lo, phi_rec, a_rec, a_j, phi_aj;
  int iz;
  // ---- result
  int nit, status;
  bool done;
};

BF_HD inline double dot(const double *a, const double *b, int n) {
  double s = 0;
  for (int i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}

// x0 [n], H0 [n, n] or nullptr (identity); maxiter <= 0: 200 n
BF_HD inline void init(Run &c, int n, const double *x0, const double *H0,
                       double gtol, double c1, double c2, double xrtol,
                       int maxiter) {
  c.n = n;
  c.pc = 0;
  c.nrows = 0;
  c.has_x = c.has_f = c.has_g = false;
  c.f = 0;
  c.nfev = c.ngev = 0;
  c.gtol = gtol, c.c1 = c1, c.c2 = c2, c.xrtol = xrtol;
  c.maxiter = maxiter > 0 ? maxiter : n * 200;
  for (int i = 0; i < n; i++) c.xk[i] = x0[i];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++)
      c.Hk[i * n + j] = H0 ? H0[i * n + j] : (i == j ? 1.0 : 0.0);
  c.fval = 0;
  c.nit = 0, c.status = 0;
  c.done = false;
}

// ---- ScalarFunction -------------------------------------------------------
BF_HD inline void sf_set_x(Run &c, const double *x) {
  bool same = c.has_x;
  if (same)
    for (int i = 0; i < c.n; i++)
      if (!(x[i] == c.sx[i])) same = false;
  if (!same) {
    for (int i = 0; i < c.n; i++) c.sx[i] = x[i];
    c.has_x = true;
    c.has_f = c.has_g = false;
  }
}
BF_HD inline void fd_points(const Run &c, double *out) {  // [n, n]
  const int n = c.n;
  bool anyzero = false;
  for (int j = 0; j < n; j++)
    if ((c.sx[j] + EPS_FD) - c.sx[j] == 0) anyzero = true;
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) out[i * n + j] = c.sx[j];
    const double dx = (c.sx[i] + EPS_FD) - c.sx[i];
    double h = EPS_FD;
    if (anyzero && dx == 0)
      h = EPS_FD * (c.sx[i] >= 0 ? 1.0 : -1.0) *
          std::fmax(1.0, std::fabs(c.sx[i]));
    out[i * n + i] = c.sx[i] + h;
  }
}
BF_HD inline void req_f(Run &c) {
  for (int i = 0; i < c.n; i++) c.rows[i] = c.sx[i];
  c.nrows = 1;
}
BF_HD inline void req_g(Run &c) {
  fd_points(c, c.rows);
  c.nrows = c.n;
}
BF_HD inline void req_fg(Run &c) {
  for (int i = 0; i < c.n; i++) c.rows[i] = c.sx[i];
  fd_points(c, c.rows + c.n);
  c.nrows = c.n + 1;
}
BF_HD inline void fin_f(Run &c) {
  c.f = c.vals[0];
  c.has_f = true;
  c.nfev += 1;
}
BF_HD inline void fin_g(Run &c, const double *f1, const double *x1) {
  for (int i = 0; i < c.n; i++)
    c.g[i] = (f1[i] - c.f) / (x1[i * c.n + i] - c.sx[i]);
  c.has_g = true;
  c.nfev += c.n;
  c.ngev += 1;
}

#define BF_YIELD_(id)  \
  do {                 \
    c.pc = (id) + 1;   \
    return;            \
    case (id) + 1:;    \
  } while (0)
#define BF_YIELD() BF_YIELD_(__COUNTER__)

// ScalarFunction.fun / .grad / (fun, grad) at xv; the value (where there is one)
// is c.f afterwards, the gradient c.g
#define BF_SF_FUN(xv)     \
  do {                    \
    sf_set_x(c, xv);      \
    if (!c.has_f) {       \
      req_f(c);           \
      BF_YIELD();         \
      fin_f(c);           \
    }                     \
  } while (0)
#define BF_SF_GRAD(xv)              \
  do {                              \
    sf_set_x(c, xv);                \
    if (!c.has_g) {                 \
      if (!c.has_f) {               \
        req_f(c);                   \
        BF_YIELD();                 \
        fin_f(c);                   \
      }                             \
      req_g(c);                     \
      BF_YIELD();                   \
      fin_g(c, c.vals, c.rows);     \
    }                               \
  } while (0)
#define BF_SF_FUN_GRAD(xv)                \
  do {                                    \
    sf_set_x(c, xv);                      \
    if (!c.has_f && !c.has_g) {           \
      req_fg(c);                          \
      BF_YIELD();                         \
      fin_f(c);                           \
      fin_g(c, c.vals + 1, c.rows + c.n); \
    } else {                              \
      BF_SF_FUN(xv);                      \
      BF_SF_GRAD(xv);                     \
    }                                     \
  } while (0)

// Runs `c` to its next request (c.nrows > 0, c.rows) or to its end (c.done).
// Before every call but the first the caller has written c.vals[0 .. nrows).
BF_HD inline void advance(Run &c) {
  const int n = c.n;
  c.nrows = 0;
  switch (c.pc) {
    case 0:;
      BF_SF_FUN_GRAD(c.xk);
      c.old_fval = c.f;
      for (int i = 0; i < n; i++) c.gfk[i] = c.g[i];
      c.k = 0, c.warnflag = 0;
      // np.linalg.norm: sqrt of the sum of squares
      c.old_old_fval = c.old_fval + std::sqrt(dot(c.gfk, c.gfk, n)) / 2;
      c.have_old_old = true;
      c.gnorm = 0;
      for (int i = 0; i < n; i++) {
        const double a = std::fabs(c.gfk[i]);
        if (a > c.gnorm || a != a) c.gnorm = a;  // np.amax propagates nan
      }
      while (c.gnorm > c.gtol && c.k < c.maxiter) {
        for (int i = 0; i < n; i++) c.pk[i] = -dot(c.Hk + i * n, c.gfk, n);
        c.alpha_k = 0, c.fval = 0, c.ofv = 0;
        c.have_stp = false, c.have_gnew = false;
        // ---------------- line_search_wolfe1 (amin=1e-100, amax=1e100) ----------
        c.derphi0 = dot(c.gfk, c.pk, n);
        c.phi0 = c.old_fval;
        if (c.have_old_old && c.derphi0 != 0) {
          c.alpha1 = 1.01 * 2 * (c.phi0 - c.old_old_fval) / c.derphi0;
          c.alpha1 = (c.alpha1 < 1.0) ? c.alpha1 : 1.0;  // min(1.0, alpha1)
          if (c.alpha1 < 0) c.alpha1 = 1.0;
        } else {
          c.alpha1 = 1.0;
        }
        c.ds.reset(c.c1, c.c2, 1e-14, 1e-100, 1e100);
        c.phi1 = c.phi0, c.derphi1 = c.derphi0, c.stp = c.alpha1;
        for (int i = 0; i < n; i++) c.gfkp1[i] = c.gfk[i];
        c.task = T_FG;
        c.stp_ok = false;
        for (c.it = 0; c.it < 100; c.it++) {
          c.stp = c.alpha1;
          c.task = c.ds.step(c.stp, c.phi1, c.derphi1);
          if (!std::isfinite(c.stp)) {
            c.task = T_WARN;
            c.stp_ok = false;
            break;
          }
          c.stp_ok = true;
          if (c.task == T_FG) {
            c.alpha1 = c.stp;
            for (int i = 0; i < n; i++) c.xt[i] = c.xk[i] + c.stp * c.pk[i];
            BF_SF_FUN_GRAD(c.xt);
            c.phi1 = c.f;
            for (int i = 0; i < n; i++) c.gfkp1[i] = c.g[i];
            c.derphi1 = dot(c.gfkp1, c.pk, n);
          } else {
            break;
          }
        }
        if (c.it == 100) {
          c.stp_ok = false;
          c.task = T_WARN;
        }
        if (c.task == T_ERROR || c.task == T_WARN) c.stp_ok = false;
        if (c.stp_ok) {
          c.have_stp = true;
          c.alpha_k = c.stp;
          c.fval = c.phi1;
          c.ofv = c.phi0;
          c.have_gnew = true;
        }
        // ---------------- line_search_wolfe2 fall-back ---------------------------
        if (!c.have_stp) {
          c.derphi0 = dot(c.gfk, c.pk, n);
          c.phi0 = c.old_fval;
          c.old_phi0 = c.old_old_fval;
          c.alpha0 = 0;
          if (c.have_old_old && c.derphi0 != 0) {
            c.alpha1 = 1.01 * 2 * (c.phi0 - c.old_phi0) / c.derphi0;
            c.alpha1 = (c.alpha1 < 1.0) ? c.alpha1 : 1.0;
          } else {
            c.alpha1 = 1.0;
          }
          if (c.alpha1 < 0) c.alpha1 = 1.0;
          c.alpha1 = (1e100 < c.alpha1) ? 1e100 : c.alpha1;  // amax
          for (int i = 0; i < n; i++) c.xt[i] = c.xk[i] + c.alpha1 * c.pk[i];
          BF_SF_FUN(c.xt);
          c.phi_a1 = c.f;
          c.phi_a0 = c.phi0, c.derphi_a0 = c.derphi0;
          c.star_alpha = false, c.star_der = false;
          c.alpha_star = 0, c.phi_star = 0;
          c.do_zoom = false;
          c.z_lo = 0, c.z_hi = 0, c.zphi_lo = 0, c.zphi_hi = 0, c.zder_lo = 0;
          c.fell_through = true;
          for (c.i2 = 0; c.i2 < 10; c.i2++) {
            if (c.alpha1 == 0 || c.alpha0 > 1e100) {
              c.star_alpha = false;
              c.phi_star = c.phi0;
              c.star_der = false;
              c.phi0 = c.old_phi0;
              c.fell_through = false;
              break;
            }
            if ((c.phi_a1 > c.phi0 + c.c1 * c.alpha1 * c.derphi0) ||
                ((c.phi_a1 >= c.phi_a0) && c.i2 > 0)) {
              c.do_zoom = true;
              c.z_lo = c.alpha0, c.z_hi = c.alpha1, c.zphi_lo = c.phi_a0,
              c.zphi_hi = c.phi_a1, c.zder_lo = c.derphi_a0;
              c.fell_through = false;
              break;
            }
            for (int q = 0; q < n; q++) c.xt[q] = c.xk[q] + c.alpha1 * c.pk[q];
            BF_SF_GRAD(c.xt);
            for (int q = 0; q < n; q++) c.gfkp1[q] = c.g[q];
            c.derphi_a1 = dot(c.gfkp1, c.pk, n);
            if (std::fabs(c.derphi_a1) <= -c.c2 * c.derphi0) {
              c.star_alpha = true;
              c.alpha_star = c.alpha1;
              c.phi_star = c.phi_a1;
              c.star_der = true;
              c.fell_through = false;
              break;
            }
            if (c.derphi_a1 >= 0) {
              c.do_zoom = true;
              c.z_lo = c.alpha1, c.z_hi = c.alpha0, c.zphi_lo = c.phi_a1,
              c.zphi_hi = c.phi_a0, c.zder_lo = c.derphi_a1;
              c.fell_through = false;
              break;
            }
            {
              double alpha2 = 2 * c.alpha1;
              alpha2 = (1e100 < alpha2) ? 1e100 : alpha2;
              c.alpha0 = c.alpha1;
              c.alpha1 = alpha2;
            }
            c.phi_a0 = c.phi_a1;
            for (int q = 0; q < n; q++) c.xt[q] = c.xk[q] + c.alpha1 * c.pk[q];
            BF_SF_FUN(c.xt);
            c.phi_a1 = c.f;
            c.derphi_a0 = c.derphi_a1;
          }
          if (c.fell_through) {  // the for-else of scalar_search_wolfe2
            c.star_alpha = true;
            c.alpha_star = c.alpha1;
            c.phi_star = c.phi_a1;
            c.star_der = false;
          }
          if (c.do_zoom) {
            c.a_lo = c.z_lo, c.a_hi = c.z_hi, c.phi_lo = c.zphi_lo,
            c.phi_hi = c.zphi_hi, c.derphi_lo = c.zder_lo;
            c.iz = 0;
            c.phi_rec = c.phi0, c.a_rec = 0;
            c.star_alpha = false;
            c.star_der = false;
            while (true) {
              {
                const double delta1 = 0.2, delta2 = 0.1;
                const double dalpha = c.a_hi - c.a_lo;
                double a, b;
                if (dalpha < 0)
                  a = c.a_hi, b = c.a_lo;
                else
                  a = c.a_lo, b = c.a_hi;
                double a_j = 0, cchk = 0;
                bool have_aj = false;
                if (c.iz > 0) {
                  cchk = delta1 * dalpha;
                  have_aj = cubicmin(c.a_lo, c.phi_lo, c.derphi_lo, c.a_hi, c.phi_hi,
                                     c.a_rec, c.phi_rec, a_j);
                }
                if (c.iz == 0 || !have_aj || a_j > b - cchk || a_j < a + cchk) {
                  const double qchk = delta2 * dalpha;
                  have_aj =
                      quadmin(c.a_lo, c.phi_lo, c.derphi_lo, c.a_hi, c.phi_hi, a_j);
                  if (!have_aj || a_j > b - qchk || a_j < a + qchk)
                    a_j = c.a_lo + 0.5 * dalpha;
                }
                c.a_j = a_j;
              }
              for (int q = 0; q < n; q++) c.xt[q] = c.xk[q] + c.a_j * c.pk[q];
              BF_SF_FUN(c.xt);
              c.phi_aj = c.f;
              if ((c.phi_aj > c.phi0 + c.c1 * c.a_j * c.derphi0) ||
                  (c.phi_aj >= c.phi_lo)) {
                c.phi_rec = c.phi_hi, c.a_rec = c.a_hi;
                c.a_hi = c.a_j, c.phi_hi = c.phi_aj;
              } else {
                for (int q = 0; q < n; q++) c.xt[q] = c.xk[q] + c.a_j * c.pk[q];
                BF_SF_GRAD(c.xt);
                for (int q = 0; q < n; q++) c.gfkp1[q] = c.g[q];
                {
                  const double derphi_aj = dot(c.gfkp1, c.pk, n);
                  if (std::fabs(derphi_aj) <= -c.c2 * c.derphi0) {
                    c.star_alpha = true;
                    c.alpha_star = c.a_j;
                    c.phi_star = c.phi_aj;
                    c.star_der = true;
                    break;
                  }
                  if (derphi_aj * (c.a_hi - c.a_lo) >= 0) {
                    c.phi_rec = c.phi_hi, c.a_rec = c.a_hi;
                    c.a_hi = c.a_lo, c.phi_hi = c.phi_lo;
                  } else {
                    c.phi_rec = c.phi_lo, c.a_rec = c.a_lo;
                  }
                  c.a_lo = c.a_j, c.phi_lo = c.phi_aj, c.derphi_lo = derphi_aj;
                }
              }
              c.iz += 1;
              if (c.iz > 10) break;  // (None, None, None)
            }
          }
          if (c.star_alpha) {
            c.have_stp = true;
            c.alpha_k = c.alpha_star;
            c.fval = c.phi_star;
            c.ofv = c.phi0;
            c.have_gnew = c.star_der;  // gval[0] of the last derphi call
          }
        }
        if (!c.have_stp) {
          c.warnflag = 2;
          break;
        }
        c.old_fval = c.fval;
        c.old_old_fval = c.ofv;
        c.have_old_old = true;
        for (int i = 0; i < n; i++) {
          c.sk[i] = c.alpha_k * c.pk[i];
          c.xk[i] = c.xk[i] + c.sk[i];
        }
        if (!c.have_gnew) {
          BF_SF_GRAD(c.xk);
          for (int i = 0; i < n; i++) c.gfkp1[i] = c.g[i];
        }
        for (int i = 0; i < n; i++) {
          c.yk[i] = c.gfkp1[i] - c.gfk[i];
          c.gfk[i] = c.gfkp1[i];
        }
        c.k += 1;
        c.gnorm = 0;
        for (int i = 0; i < n; i++) {
          const double a = std::fabs(c.gfk[i]);
          if (a > c.gnorm || a != a) c.gnorm = a;
        }
        if (c.gnorm <= c.gtol) break;
        {
          double pp = 0, xx = 0;
          for (int i = 0; i < n; i++) pp += c.pk[i] * c.pk[i];
          for (int i = 0; i < n; i++) xx += c.xk[i] * c.xk[i];
          if (c.alpha_k * std::sqrt(pp) <= c.xrtol * (c.xrtol + std::sqrt(xx)))
            break;
        }
        if (!std::isfinite(c.old_fval)) {
          c.warnflag = 2;
          break;
        }
        {
          const double rhok_inv = dot(c.yk, c.sk, n);
          const double rhok = (rhok_inv == 0.) ? 1000.0 : 1. / rhok_inv;
          // Hk = A1 Hk A2 + rhok sk sk^T, A1 = I - sk yk^T rhok, A2 = I - yk sk^T rhok
          // (the entries of A1 / A2 formed where they are used: the same products)
          for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
              double s = 0;
              for (int q = 0; q < n; q++)
                s += c.Hk[i * n + q] *
                     ((q == j ? 1.0 : 0.0) - c.yk[q] * c.sk[j] * rhok);
              c.T1[i * n + j] = s;
            }
          for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
              double s = 0;
              for (int q = 0; q < n; q++)
                s += ((i == q ? 1.0 : 0.0) - c.sk[i] * c.yk[q] * rhok) *
                     c.T1[q * n + j];
              c.Hk[i * n + j] = s + (rhok * c.sk[i]) * c.sk[j];
            }
        }
      }
      c.fval = c.old_fval;
      if (c.warnflag == 2) {
      } else if (c.k >= c.maxiter) {
        c.warnflag = 1;
      } else {
        bool xnan = false;
        for (int i = 0; i < n; i++)
          if (c.xk[i] != c.xk[i]) xnan = true;
        if (c.gnorm != c.gnorm || c.fval != c.fval || xnan) c.warnflag = 3;
      }
      c.nit = c.k;
      c.status = c.warnflag;
      c.nrows = 0;
      c.done = true;
      c.pc = -1;
      return;
    default:
      return;  // finished runs stay finished
  }
}

#undef BF_SF_FUN
#undef BF_SF_GRAD
#undef BF_SF_FUN_GRAD
#undef BF_YIELD
#undef BF_YIELD_

}  // namespace rvs_bfgs
