// Device-resident lock-step Nelder-Mead for S independent simplices, and the
// parameter mapping of vel_fit.chisq_func around the batched objective.
//
// vel_fit.process runs scipy's Nelder-Mead once per spectrum (vel_fit.py:627-637);
// here the S state machines live in HBM and advance together.  One round is
//   rvs_nm_begin   termination test, centroid, reflection point  -> list1, X1
//   <objective>    F1 = f(X1)
//   rvs_nm_decide  branch per simplex; expansion / contraction point -> list2, X2
//   <objective>    F2 = f(X2)
//   rvs_nm_update  accept / replace / order; a simplex that must shrink is parked
//   (rare: rvs_nm_collect -> list3, then per vertex rvs_nm_shrink_point +
//    <objective> + rvs_nm_shrink_store, whenever the host next looks)
// Every kernel reads its job count from DEVICE memory (`counts`), so the host
// never has to synchronise inside a round: it launches with the last count it
// has seen as an upper bound (active sets only shrink) and refreshes that bound
// every few rounds.  List entries beyond the live count are padded with a copy
// of entry 0, so the objective kernels in between can run the full bound.
//
// The branch structure, constants (rho=1, chi=2, psi=0.5, sigma=0.5), stable
// vertex ordering (NaN last) and the order of the floating-point operations
// follow scipy/optimize/_optimize.py::_minimize_neldermead, the same as
// tests/refmachines/neldermead_torch.py (which tests/ checks against scipy itself).
#include "common.h"
#include "objective_sum.h"
#include "nm_internal.h"

// numpy evaluates every product and sum of the simplex arithmetic separately;
// a fused multiply-add would change the last bit and, eventually, the path
#pragma clang fp contract(off)

#define NM_MAXN 8
#define NM_NT 1024
// the one-block kernels that hold a whole simplex in registers ((N + 1) N + N + 1
// doubles: 98 VGPRs at N = 6, 162 at N = 8): eight waves, so that a wave has 256
#ifndef NM_UNT
#define NM_UNT 512
#endif

// counts layout (int32[8]): [0] jobs of list1, [1] jobs of list2, [2] length
// of list3, [3] simplices stepping this round, [4] simplices parked for a shrink
// flags: bit 0 active, bit 1 converged (success), bit 2 shrink pending
template <int NT = NM_NT>
__device__ __forceinline__ int block_excl_scan(int flag, int *total,
                                               int *sh /*[NT/64 + 1]*/) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned long long m = __ballot(flag);
  const int pre = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) sh[w] = __popcll(m);
  __syncthreads();
  int off = 0, tot = 0;
  for (int i = 0; i < NT / 64; i++) {
    const int c = sh[i];
    if (i < w) off += c;
    tot += c;
  }
  __syncthreads();
  *total = tot;
  return off + pre;
}

// stable insertion sort of the N+1 vertices by f (NaN last), np.argsort order.
// The simplex is pulled into registers first (N is a template parameter, all
// loops unroll): sorting it in place in global memory is a chain of ~50
// dependent loads/stores, which made the bookkeeping kernels latency-bound.
template <int N>
__device__ __forceinline__ void nm_load_regs(const double *__restrict__ gs,
                                             const double *__restrict__ gf,
                                             double (&s)[N + 1][N], double (&f)[N + 1]) {
#pragma unroll
  for (int a = 0; a <= N; a++) {
    f[a] = gf[a];
#pragma unroll
    for (int i = 0; i < N; i++) s[a][i] = gs[a * N + i];
  }
}

template <int N>
__device__ __forceinline__ void nm_store_regs(double *gs, double *gf,
                                              const double (&s)[N + 1][N],
                                              const double (&f)[N + 1]) {
#pragma unroll
  for (int a = 0; a <= N; a++) {
    gf[a] = f[a];
#pragma unroll
    for (int i = 0; i < N; i++) gs[a * N + i] = s[a][i];
  }
}

template <int N>
__device__ __forceinline__ void nm_sort_regs(double (&s)[N + 1][N], double (&f)[N + 1]) {
#pragma unroll
  for (int a = 1; a <= N; a++) {
    const double fa = f[a];
    const double ka = (fa != fa) ? __builtin_inf() : fa;
    double xa[N];
#pragma unroll
    for (int i = 0; i < N; i++) xa[i] = s[a][i];
    bool placed = false;
#pragma unroll
    for (int b = a - 1; b >= 0; b--) {
      if (!placed) {
        const double fb = f[b];
        const double kb = (fb != fb) ? __builtin_inf() : fb;
        if (kb > ka) {
          f[b + 1] = fb;
#pragma unroll
          for (int i = 0; i < N; i++) s[b + 1][i] = s[b][i];
          if (b == 0) {
            f[0] = fa;
#pragma unroll
            for (int i = 0; i < N; i++) s[0][i] = xa[i];
            placed = true;
          }
        } else {
          f[b + 1] = fa;
#pragma unroll
          for (int i = 0; i < N; i++) s[b + 1][i] = xa[i];
          placed = true;
        }
      }
    }
  }
}

template <int N>
__device__ void nm_order_t(double *gsim, double *gf) {
  double s[N + 1][N], f[N + 1];
  nm_load_regs<N>(gsim, gf, s, f);
  nm_sort_regs<N>(s, f);
  nm_store_regs<N>(gsim, gf, s, f);
}

__device__ void nm_order(double *sim, double *f, int N) {
  switch (N) {
    case 1: nm_order_t<1>(sim, f); break;
    case 2: nm_order_t<2>(sim, f); break;
    case 3: nm_order_t<3>(sim, f); break;
    case 4: nm_order_t<4>(sim, f); break;
    case 5: nm_order_t<5>(sim, f); break;
    case 6: nm_order_t<6>(sim, f); break;
    case 7: nm_order_t<7>(sim, f); break;
    default: nm_order_t<8>(sim, f); break;
  }
}

// termination test and reflection point of one simplex, all loads up front
// (N compile-time); returns 1 when converged
template <int N>
__device__ __forceinline__ int nm_test_regs(const double (&s)[N + 1][N],
                                            const double (&f)[N + 1], double xatol,
                                            double fatol, double *xr) {
  double dx = 0, df = 0;
  bool anynan = (f[0] != f[0]);
#pragma unroll
  for (int k = 1; k <= N; k++) {
#pragma unroll
    for (int i = 0; i < N; i++) {
      dx = fmax(dx, fabs(s[k][i] - s[0][i]));
      if (s[k][i] != s[k][i] || s[0][i] != s[0][i]) anynan = true;
    }
    df = fmax(df, fabs(f[0] - f[k]));
    if (f[k] != f[k]) anynan = true;
  }
  // NaN propagates like np.max: a NaN difference never passes the test
  if (!anynan && dx <= xatol && df <= fatol) return 1;
#pragma unroll
  for (int i = 0; i < N; i++) {
    double xb = s[0][i];
#pragma unroll
    for (int k = 1; k < N; k++) xb = xb + s[k][i];
    xb = xb / N;
    xr[i] = (1 + 1.0) * xb - 1.0 * s[N][i];
  }
  return 0;
}

template <int N>
__device__ __forceinline__ int nm_begin_row(const double *__restrict__ gs,
                                            const double *__restrict__ gf,
                                            double xatol, double fatol,
                                            double *xr) {
  double s[N + 1][N], f[N + 1];
  nm_load_regs<N>(gs, gf, s, f);
  return nm_test_regs<N>(s, f, xatol, fatol, xr);
}

__global__ void __launch_bounds__(NM_NT)
    nm_begin_kernel(int S, int N, double xatol, double fatol, int maxiter,
                    const double *__restrict__ sim,
                    const double *__restrict__ fsim,
                    const int32_t *__restrict__ nit, int32_t *__restrict__ flags,
                    int32_t *__restrict__ list1, double *__restrict__ X1,
                    int32_t *__restrict__ counts, int jbound) {
  __shared__ int sh[NM_NT / 64 + 1];
  int base_out = 0;
  for (int r0 = 0; r0 < S; r0 += NM_NT) {
    const int r = r0 + threadIdx.x;
    int go = 0;
    double xr[NM_MAXN];
    if (r < S && (flags[r] & 5) == 1) {  // active and no shrink pending
      const double *s = sim + (int64_t)r * (N + 1) * N;
      const double *f = fsim + (int64_t)r * (N + 1);
      if (nit[r] >= maxiter) {
        flags[r] &= ~1;  // scipy: while-condition fails -> warnflag 2
      } else {
        int conv = 0;
        switch (N) {
          case 1: conv = nm_begin_row<1>(s, f, xatol, fatol, xr); break;
          case 2: conv = nm_begin_row<2>(s, f, xatol, fatol, xr); break;
          case 3: conv = nm_begin_row<3>(s, f, xatol, fatol, xr); break;
          case 4: conv = nm_begin_row<4>(s, f, xatol, fatol, xr); break;
          case 5: conv = nm_begin_row<5>(s, f, xatol, fatol, xr); break;
          case 6: conv = nm_begin_row<6>(s, f, xatol, fatol, xr); break;
          case 7: conv = nm_begin_row<7>(s, f, xatol, fatol, xr); break;
          default: conv = nm_begin_row<8>(s, f, xatol, fatol, xr); break;
        }
        if (conv)
          flags[r] = (flags[r] & ~1) | 2;  // converged: success
        else
          go = 1;
      }
    }
    int tot;
    const int pos = base_out + block_excl_scan(go, &tot, sh);
    if (go) {
      list1[pos] = r;
      for (int i = 0; i < N; i++) X1[(int64_t)pos * N + i] = xr[i];
    }
    base_out += tot;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    counts[0] = base_out;
    counts[3] = base_out;
  }
  // pad the unused tail with a copy of entry 0 (or simplex 0's best vertex)
  if (base_out == 0 && threadIdx.x == 0) {
    list1[0] = 0;
    for (int i = 0; i < N; i++) X1[i] = sim[i];
  }
  __syncthreads();
  for (int j = max(base_out, 1) + threadIdx.x; j < jbound; j += NM_NT) {
    list1[j] = list1[0];
    for (int i = 0; i < N; i++) X1[(int64_t)j * N + i] = X1[i];
  }
}

// cases: 0 accept reflection, 1 expansion, 2 outside contraction, 3 inside
__global__ void __launch_bounds__(NM_NT)
    nm_decide_kernel(int N, const double *__restrict__ sim,
                     const double *__restrict__ fsim,
                     const int32_t *__restrict__ list1,
                     const double *__restrict__ F1, int32_t *__restrict__ cases,
                     int32_t *__restrict__ pos2, int32_t *__restrict__ list2,
                     double *__restrict__ X2, int32_t *__restrict__ counts,
                     int jbound) {
  __shared__ int sh[NM_NT / 64 + 1];
  const int J = min(counts[0], jbound);
  int base_out = 0;
  for (int j0 = 0; j0 < J; j0 += NM_NT) {
    const int j = j0 + threadIdx.x;
    int go = 0;
    double x2[NM_MAXN];
    int r = 0;
    if (j < J) {
      r = list1[j];
      const double *s = sim + (int64_t)r * (N + 1) * N;
      const double *f = fsim + (int64_t)r * (N + 1);
      const double fxr = F1[j];
      int c;
      if (fxr < f[0])
        c = 1;
      else if (fxr < f[N - 1])
        c = 0;
      else if (fxr < f[N])
        c = 2;
      else
        c = 3;
      cases[j] = c;
      if (c != 0) {
        go = 1;
        for (int i = 0; i < N; i++) {
          double xb = s[i];
          for (int k = 1; k < N; k++) xb = xb + s[k * N + i];
          xb = xb / N;
          const double w = s[N * N + i];
          if (c == 1)
            x2[i] = (1 + 1.0 * 2.0) * xb - 1.0 * 2.0 * w;
          else if (c == 2)
            x2[i] = (1 + 0.5 * 1.0) * xb - 0.5 * 1.0 * w;
          else
            x2[i] = (1 - 0.5) * xb + 0.5 * w;
        }
      }
    }
    int tot;
    const int pos = base_out + block_excl_scan(go, &tot, sh);
    if (j < J) pos2[j] = go ? pos : -1;
    if (go) {
      list2[pos] = r;
      for (int i = 0; i < N; i++) X2[(int64_t)pos * N + i] = x2[i];
    }
    base_out += tot;
  }
  __syncthreads();
  if (threadIdx.x == 0) counts[1] = base_out;
  if (base_out == 0 && threadIdx.x == 0) {
    list2[0] = 0;
    for (int i = 0; i < N; i++) X2[i] = sim[i];
  }
  __syncthreads();
  for (int j = max(base_out, 1) + threadIdx.x; j < jbound; j += NM_NT) {
    list2[j] = list2[0];
    for (int i = 0; i < N; i++) X2[(int64_t)j * N + i] = X2[i];
  }
}

__global__ void __launch_bounds__(NM_NT)
    nm_update_kernel(int N, double *__restrict__ sim, double *__restrict__ fsim,
                     int32_t *__restrict__ nit, int32_t *__restrict__ nfev,
                     const int32_t *__restrict__ list1,
                     const double *__restrict__ X1,
                     const double *__restrict__ F1,
                     const int32_t *__restrict__ cases,
                     const int32_t *__restrict__ pos2,
                     const double *__restrict__ X2,
                     const double *__restrict__ F2, int32_t *__restrict__ flags,
                     int32_t *__restrict__ counts, int jbound) {
  __shared__ int sh[NM_NT / 64 + 1];
  const int J = min(counts[0], jbound);
  int base_out = 0;
  for (int j0 = 0; j0 < J; j0 += NM_NT) {
    const int j = j0 + threadIdx.x;
    int go = 0, r = 0;
    if (j < J) {
      r = list1[j];
      double *s = sim + (int64_t)r * (N + 1) * N;
      double *f = fsim + (int64_t)r * (N + 1);
      const int c = cases[j];
      const double fxr = F1[j];
      const int p2 = pos2[j];
      const double f2 = (p2 >= 0) ? F2[p2] : __builtin_inf();
      bool take2 = false, taker = false;
      if (c == 0)
        taker = true;
      else if (c == 1) {
        if (f2 < fxr)
          take2 = true;
        else
          taker = true;
      } else if (c == 2)
        take2 = (f2 <= fxr);
      else
        take2 = (f2 < f[N]);
      nfev[r] += (c == 0) ? 1 : 2;
      if (take2 || taker) {
        const double *src = take2 ? (X2 + (int64_t)p2 * N) : (X1 + (int64_t)j * N);
        for (int i = 0; i < N; i++) s[N * N + i] = src[i];
        f[N] = take2 ? f2 : fxr;
        nm_order(s, f, N);
        nit[r] += 1;
      } else {
        go = 1;  // shrink: parked (flag bit 2) until the host runs the shrink
      }
    }
    int tot;
    const int pos = base_out + block_excl_scan(go, &tot, sh);
    if (go) flags[r] |= 4;
    (void)pos;
    base_out += tot;
  }
  __syncthreads();
  if (threadIdx.x == 0) counts[4] += base_out;  // simplices waiting to shrink
}

// list3 = simplices with a pending shrink (flag bit 2), counts[2] = how many
__global__ void __launch_bounds__(NM_NT)
    nm_collect_kernel(int S, const int32_t *__restrict__ flags,
                      int32_t *__restrict__ list3,
                      int32_t *__restrict__ counts) {
  __shared__ int sh[NM_NT / 64 + 1];
  int base_out = 0;
  for (int r0 = 0; r0 < S; r0 += NM_NT) {
    const int r = r0 + threadIdx.x;
    const int go = (r < S && (flags[r] & 4)) ? 1 : 0;
    int tot;
    const int pos = base_out + block_excl_scan(go, &tot, sh);
    if (go) list3[pos] = r;
    base_out += tot;
  }
  __syncthreads();
  if (threadIdx.x == 0) counts[2] = base_out;
}

// vertex k (1..N) of every shrinking simplex: sim[k] = sim[0] + 0.5 (sim[k]-sim[0])
__global__ void __launch_bounds__(256)
    nm_shrink_point_kernel(int N, int k, double *__restrict__ sim,
                           const int32_t *__restrict__ list3,
                           double *__restrict__ X3,
                           const int32_t *__restrict__ counts, int jbound) {
  const int J = min(counts[2], jbound);
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= jbound) return;
  const int jj = (j < J) ? j : 0;
  const int r = (J > 0) ? list3[jj] : 0;
  double *s = sim + (int64_t)r * (N + 1) * N;
  for (int i = 0; i < N; i++) {
    double v = s[i];
    if (J > 0) {
      v = s[i] + 0.5 * (s[k * N + i] - s[i]);
      if (j < J) s[k * N + i] = v;
    }
    X3[(int64_t)j * N + i] = v;
  }
}

__global__ void __launch_bounds__(256)
    nm_shrink_store_kernel(int N, int k, double *__restrict__ sim,
                           double *__restrict__ fsim,
                           int32_t *__restrict__ nit, int32_t *__restrict__ nfev,
                           int32_t *__restrict__ flags,
                           const int32_t *__restrict__ list3,
                           const double *__restrict__ F3,
                           int32_t *__restrict__ counts, int jbound) {
  const int J = min(counts[2], jbound);
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= J) return;
  const int r = list3[j];
  fsim[(int64_t)r * (N + 1) + k] = F3[j];
  if (k == N) {
    nm_order(sim + (int64_t)r * (N + 1) * N, fsim + (int64_t)r * (N + 1), N);
    nit[r] += 1;
    nfev[r] += N;
    flags[r] &= ~4;
    if (j == 0) counts[4] = 0;
  }
}

extern "C" int rvs_nm_begin(int S, int N, double xatol, double fatol,
                            int maxiter, const double *sim, const double *fsim,
                            const int32_t *nit, int32_t *flags, int32_t *list1,
                            double *X1, int32_t *counts, int jbound,
                            void *stream) {
  if (S < 1 || N < 1 || N > NM_MAXN || jbound < 1) return RVS_E_ARG;
  hipLaunchKernelGGL(nm_begin_kernel, dim3(1), dim3(NM_NT), 0,
                     rvs_stream(stream), S, N, xatol, fatol, maxiter, sim, fsim,
                     nit, flags, list1, X1, counts, jbound);
  RVS_LAUNCH_CHECK();
  return 0;
}

extern "C" int rvs_nm_decide(int N, const double *sim, const double *fsim,
                             const int32_t *list1, const double *F1,
                             int32_t *cases, int32_t *pos2, int32_t *list2,
                             double *X2, int32_t *counts, int jbound,
                             void *stream) {
  if (N < 1 || N > NM_MAXN || jbound < 1) return RVS_E_ARG;
  hipLaunchKernelGGL(nm_decide_kernel, dim3(1), dim3(NM_NT), 0,
                     rvs_stream(stream), N, sim, fsim, list1, F1, cases, pos2,
                     list2, X2, counts, jbound);
  RVS_LAUNCH_CHECK();
  return 0;
}

extern "C" int rvs_nm_update(int N, double *sim, double *fsim, int32_t *nit,
                             int32_t *nfev, const int32_t *list1,
                             const double *X1, const double *F1,
                             const int32_t *cases, const int32_t *pos2,
                             const double *X2, const double *F2, int32_t *flags,
                             int32_t *counts, int jbound, void *stream) {
  if (N < 1 || N > NM_MAXN || jbound < 1) return RVS_E_ARG;
  hipLaunchKernelGGL(nm_update_kernel, dim3(1), dim3(NM_NT), 0,
                     rvs_stream(stream), N, sim, fsim, nit, nfev, list1, X1, F1,
                     cases, pos2, X2, F2, flags, counts, jbound);
  RVS_LAUNCH_CHECK();
  return 0;
}

extern "C" int rvs_nm_collect(int S, const int32_t *flags, int32_t *list3,
                              int32_t *counts, void *stream) {
  if (S < 1) return RVS_E_ARG;
  hipLaunchKernelGGL(nm_collect_kernel, dim3(1), dim3(NM_NT), 0,
                     rvs_stream(stream), S, flags, list3, counts);
  RVS_LAUNCH_CHECK();
  return 0;
}

extern "C" int rvs_nm_shrink_point(int N, int k, double *sim,
                                   const int32_t *list3, double *X3,
                                   const int32_t *counts, int jbound,
                                   void *stream) {
  if (N < 1 || N > NM_MAXN || k < 1 || k > N || jbound < 1) return RVS_E_ARG;
  hipLaunchKernelGGL(nm_shrink_point_kernel, dim3((jbound + 255) / 256),
                     dim3(256), 0, rvs_stream(stream), N, k, sim, list3, X3,
                     counts, jbound);
  RVS_LAUNCH_CHECK();
  return 0;
}

extern "C" int rvs_nm_shrink_store(int N, int k, double *sim, double *fsim,
                                   int32_t *nit, int32_t *nfev, int32_t *flags,
                                   const int32_t *list3, const double *F3,
                                   int32_t *counts, int jbound, void *stream) {
  if (N < 1 || N > NM_MAXN || k < 1 || k > N || jbound < 1) return RVS_E_ARG;
  hipLaunchKernelGGL(nm_shrink_store_kernel, dim3((jbound + 255) / 256),
                     dim3(256), 0, rvs_stream(stream), N, k, sim, fsim, nit,
                     nfev, flags, list3, F3, counts, jbound);
  RVS_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
// vel_fit.ParamMapper.forward + the range / finiteness guard of chisq_func +
// VSiniMapper.to_vsini + the Normal priors of chisq_func0 (vel_fit.py:95-254)
// for J rows of the optimiser's parameter vectors.
//   X [J, n]: (vel, [vsini], free stellar parameters in specParams order)
//   src [ndim]: column of X feeding stellar parameter i, or -1 = fixed
//   fixed [S, ndim], vsini_fixed [S] (used when vsini_col < 0, nullable = no
//   rotation), prior_mean / prior_isig [S, ndim] (nullable; isig 0 = no prior)
// out: job_spec[j] = list[j], vel, vsini (nullable), params [J, ndim],
//      extra[j] = vsini penalty + priors, bad[j] (row answered with 1e30; its
//      vel/params are replaced by vel 0 / the fixed+start values `safe`)
// ---------------------------------------------------------------------------
struct MapSrc {
  int src[NM_MAXN];
};

// vel_fit.chisq_func's parameter mapping for ONE row: the optimiser's vector x of
// simplex r -> (velocity, vsini, template parameters, prior penalty, bad flag) of job j
struct MapP {
  int n, ndim, vsini_col;
  MapSrc M;
  const double *fixed, *vsini_fixed, *safe, *prior_mean, *prior_isig;
  double min_vel, max_vel, max_vsini;
  int32_t *job_spec;
  double *vel, *vsini, *params, *extra;
  int32_t *bad;
};

// entry idx of a vector that lives in registers (a chain of selects; each entry
// through a register first, or the compiler turns the chain into a load at a selected
// offset and the vector into scratch memory: template_dev.h sel_dim)
__device__ __forceinline__ double nm_pick(const double *x, int idx) {
  double v = x[0];
#pragma unroll
  for (int i = 1; i < NM_MAXN; i++) {
    double xi = x[i];
    asm volatile("" : "+v"(xi));
    v = (idx == i) ? xi : v;
  }
  return v;
}

// (x: the row in the caller's registers, NM_MAXN entries of which P.n count)
__device__ __forceinline__ void map_row_regs(const MapP &P, int j, int r,
                                             const double *x) {
  const int ndim = P.ndim;
  double v = x[0];
  double pen = 0;
  if (P.vsini) {
    double vs;
    if (P.vsini_col >= 0) {
      const double v0 = nm_pick(x, P.vsini_col);
      vs = fmin(fmax(v0, 0.0), P.max_vsini);  // np.clip
      if (v0 < 0 || v0 > P.max_vsini) pen += (vs - v0) * (vs - v0);
      if (v0 != v0) vs = v0;
    } else {
      vs = P.vsini_fixed[r];
    }
    P.vsini[j] = vs;
  }
  bool isbad = (v > P.max_vel) || (v < P.min_vel);
  double p[NM_MAXN];
#pragma unroll
  for (int i = 0; i < NM_MAXN; i++) {   // (p[] in registers: static indices)
    if (i >= ndim) break;
    p[i] = (P.M.src[i] >= 0) ? nm_pick(x, P.M.src[i]) : P.fixed[(int64_t)r * ndim + i];
    if (!(fabs(p[i]) <= 1.79e308)) isbad = true;
  }
  if (isbad) {
    v = 0;
#pragma unroll
    for (int i = 0; i < NM_MAXN; i++)
      if (i < ndim) p[i] = P.safe[(int64_t)r * ndim + i];
  }
  if (P.prior_mean)
#pragma unroll
    for (int i = 0; i < NM_MAXN; i++) {
      if (i >= ndim) break;
      const double d = (P.prior_mean[(int64_t)r * ndim + i] - p[i]) *
                       P.prior_isig[(int64_t)r * ndim + i];
      pen += d * d;
    }
  P.job_spec[j] = r;
  P.vel[j] = v;
#pragma unroll
  for (int i = 0; i < NM_MAXN; i++)
    if (i < ndim) P.params[(int64_t)j * ndim + i] = p[i];
  P.extra[j] = pen;
  P.bad[j] = isbad ? 1 : 0;
}

// row i < N of x (memory) <-> xr (registers: no indexing by a loop variable)
__device__ __forceinline__ void nm_get_row(double *xr, const double *x, int N) {
#pragma unroll
  for (int i = 0; i < NM_MAXN; i++) xr[i] = (i < N) ? x[i] : 0.0;
}
__device__ __forceinline__ void nm_put_row(double *x, const double *xr, int N) {
#pragma unroll
  for (int i = 0; i < NM_MAXN; i++)
    if (i < N) x[i] = xr[i];
}

__device__ __forceinline__ void map_row(const MapP &P, int j, int r,
                                        const double *__restrict__ x) {
  double xr[NM_MAXN];
  nm_get_row(xr, x, P.n);
  map_row_regs(P, j, r, xr);
}

__global__ void __launch_bounds__(256)
    proc_map_kernel(int J, const double *__restrict__ X,
                    const int32_t *__restrict__ list, MapP P) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= J) return;
  map_row(P, j, list[j], X + (int64_t)j * P.n);
}

__global__ void __launch_bounds__(256)
    proc_finish_kernel(int J, const int32_t *__restrict__ counts, int cidx,
                       const double *__restrict__ chi,
                       const double *__restrict__ extra,
                       const int32_t *__restrict__ bad,
                       const int32_t *__restrict__ job_spec,
                       const int32_t *__restrict__ job_status,
                       double *__restrict__ F,
                       int32_t *__restrict__ spec_status) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= J) return;
  F[j] = bad[j] ? 1e30 : chi[j] + extra[j];
  const int live = counts ? counts[cidx] : J;
  if (j < live && job_status[j] && !bad[j])
    atomicOr(&spec_status[job_spec[j]], job_status[j]);
}

extern "C" int rvs_proc_map(int J, int n, int ndim, const double *X,
                            const int32_t *list, const int32_t *src,
                            int vsini_col, const double *fixed,
                            const double *vsini_fixed, const double *safe,
                            const double *prior_mean, const double *prior_isig,
                            double min_vel, double max_vel, double max_vsini,
                            int32_t *job_spec, double *vel, double *vsini,
                            double *params, double *extra, int32_t *bad,
                            void *stream) {
  if (J < 1 || n < 1 || n > NM_MAXN || ndim < 1 || ndim > NM_MAXN)
    return RVS_E_ARG;
  MapP P;
  P.n = n, P.ndim = ndim, P.vsini_col = vsini_col;
  for (int i = 0; i < NM_MAXN; i++) P.M.src[i] = (i < ndim) ? src[i] : -1;
  P.fixed = fixed, P.vsini_fixed = vsini_fixed, P.safe = safe;
  P.prior_mean = prior_mean, P.prior_isig = prior_isig;
  P.min_vel = min_vel, P.max_vel = max_vel, P.max_vsini = max_vsini;
  P.job_spec = job_spec, P.vel = vel, P.vsini = vsini, P.params = params;
  P.extra = extra, P.bad = bad;
  hipLaunchKernelGGL(proc_map_kernel, dim3((J + 255) / 256), dim3(256), 0,
                     rvs_stream(stream), J, X, list, P);
  RVS_LAUNCH_CHECK();
  return 0;
}

extern "C" int rvs_proc_finish(int J, const int32_t *counts, int cidx,
                               const double *chi, const double *extra,
                               const int32_t *bad, const int32_t *job_spec,
                               const int32_t *job_status, double *F,
                               int32_t *spec_status, void *stream) {
  if (J < 1) return RVS_E_ARG;
  hipLaunchKernelGGL(proc_finish_kernel, dim3((J + 255) / 256), dim3(256), 0,
                     rvs_stream(stream), J, counts, cidx, chi, extra, bad,
                     job_spec, job_status, F, spec_status);
  RVS_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
// The rounds of the lock-step optimiser driven from C: what optimizer.py's
// DeviceNelderMead.minimize + ProcessObjective.eval do per round (begin -> map ->
// objective -> finish -> decide -> map -> objective -> finish -> update, the
// host looking at the counters every `sync_every` rounds, parked shrinks run at
// that look), without ~25 interpreter round trips per round.  The call blocks
// its host thread but not the interpreter (ctypes releases the GIL), so two
// optimiser instances on two streams can be driven by two Python threads.
// ---------------------------------------------------------------------------
int rvs_internal_nm_eval(const rvs_nm_objective *o, const int32_t *list,
                         const double *X, int J, const int32_t *counts, int cidx,
                         double *F, hipStream_t st) {
  int rc = rvs_proc_map(J, o->n, o->ndim, X, list, o->src, o->vsini_col,
                        o->fixed, o->vsini_fixed, o->safe, o->prior_mean,
                        o->prior_isig, o->min_vel, o->max_vel, o->max_vsini,
                        o->job_spec, o->vel, o->vsini, o->params, o->extra,
                        o->bad, st);
  if (rc) return rc;
  // rows behind the device count (simplices that finished since the host's last look,
  // simplices that need no second point this round) are not evaluated: a quarter of
  // the objective blocks of a run (tools/perf/nm_waste.py).  (Sizing the second
  // launch exactly by one more look per round: 2390-2445 against 2398 spectra/s,
  // no difference -- a block behind the count costs next to nothing.)
  const int32_t *live = counts ? counts + cidx : nullptr;
  if (o->nn) {
    // MLP libraries: the round's template rows and outside flags per arm, then
    // broadening + spline + chi^2 in one kernel (optimizer.py's from_templ path)
    const double *tp[8], *op[8];
    if (o->narm > 8) return RVS_E_ARG;
    // (one grouped launch chain for the arms.  Forked onto side streams the
    // per-arm chains overlapped inside one optimiser instance -- Nelder-Mead
    // 2.53 -> 2.36 s per 2000 spectra -- but two instances on two host threads,
    // which is how vel_fit.process runs a large batch, then took 3.5 s for 2.4)
    rc = rvs_template_nn_arms_n(o->params, J, live, o->ndim, o->narm, o->nn, st);
    if (rc) return rc;
    for (int a = 0; a < o->narm; a++) {
      tp[a] = o->nn[a].templ;
      op[a] = o->nn[a].outside;
    }
    rc = rvs_objective_from_template_n(o->arms, o->narm, o->npoly, tp, op,
                                       o->vsini, o->job_spec, J, live, o->vel,
                                       o->badchi, 1 | RVS_OBJ_STATUS_STORE,
                                       o->scratch, o->chi, o->jstatus, st);
  } else if (o->tri) {
    // Delaunay libraries: find_simplex + blend per arm, then the same kernel
    const double *tp[8], *op[8];
    if (o->narm > 8) return RVS_E_ARG;
    rc = rvs_internal_template_tri_arms_n(o->params, J, live, o->ndim, o->narm, o->tri,
                                          st);
    if (rc) return rc;
    for (int a = 0; a < o->narm; a++) {
      tp[a] = o->tri[a].templ;
      op[a] = o->tri[a].outside;
    }
    rc = rvs_objective_from_template_n(o->arms, o->narm, o->npoly, tp, op,
                                       o->vsini, o->job_spec, J, live, o->vel,
                                       o->badchi, 1 | RVS_OBJ_STATUS_STORE,
                                       o->scratch, o->chi, o->jstatus, st);
  } else {
    rc = rvs_objective_fused_n(o->arms, o->narm, o->npoly, o->params, o->vsini,
                               o->job_spec, J, live, o->vel, o->badchi,
                               1 | RVS_OBJ_STATUS_STORE, o->scratch, o->chi,
                               o->jstatus, st);
  }
  if (rc) return rc;
  return rvs_proc_finish(J, counts, cidx, o->chi, o->extra, o->bad, o->job_spec,
                         o->jstatus, F, o->status, st);
}


// ---------------------------------------------------------------------------
// The round as rvs_nm_run launches it: THREE launches per function evaluation --
// cell search, job order, objective kernel -- and one bookkeeping kernel between two
// evaluations, where the chain above takes seven (begin | decide | update, map,
// cell search, order, objective, sum over the arms, finish).  Between two objective
// kernels of a stream nothing else runs, and late in a run -- a few dozen simplices
// left, an objective launch of 40 us -- the chain WAS the round: 141 us of small
// launches per evaluation (tools/perf/trace_rounds.py).  One block does, for all
// rows, what the separate kernels did (same device functions, same order of the
// arithmetic, lists in the same order):
//   nm_glue_begin     termination test + reflection point of every running simplex,
//                     list1 / X1, and the parameter mapping of those rows
//   nm_glue_decide    F1 = sum over the arms + priors of the rows just evaluated;
//                     branch per simplex; second point -> list2 / X2 + their mapping
//   nm_glue_update    F2 likewise; accept / replace / order / park; then the next
//                     round's termination test + reflection point of the simplices
//                     that keep running (list1 compacted in place) + their mapping
// The function values are formed for ALL rows before any mapped row is written (the
// mapping of the next rows lives in the buffers the evaluated rows are read from).
// ---------------------------------------------------------------------------
struct NmGlue {
  rvs_nm_state m;
  MapP P;
  ObjArmOut AO;           // per-arm results of the evaluation just done
  double badchi, xatol, fatol;
  const double *pen_scale;
  int32_t *spec_status;
  int maxiter;
};

// F of row j of the evaluation just done (objective_sum_kernel + proc_finish_kernel)
__device__ __forceinline__ double glue_value(const NmGlue &G, int j) {
  const int r = G.P.job_spec[j];
  double bc = G.badchi;
  if (G.pen_scale) bc *= G.pen_scale[r];
  double tot;
  int st;
  obj_sum_row(G.AO, j, bc, 1, tot, st);
  const int isbad = G.P.bad[j];
  if (st && !isbad) atomicOr(&G.spec_status[r], st);
  return isbad ? 1e30 : tot + G.P.extra[j];
}

// nm_begin_kernel's test of simplex r: 1 = it steps this round (xr = reflection point)
__device__ __forceinline__ int glue_begin_row(const NmGlue &G, int r, double *xr) {
  int32_t *flags = G.m.flags;
  const int N = G.m.N;
  if ((flags[r] & 5) != 1) return 0;  // not active, or a shrink pending
  if (G.m.nit[r] >= G.maxiter) {
    flags[r] &= ~1;  // scipy: while-condition fails -> warnflag 2
    return 0;
  }
  const double *s = G.m.sim + (int64_t)r * (N + 1) * N;
  const double *f = G.m.fsim + (int64_t)r * (N + 1);
  int conv = 0;
  switch (N) {
    case 1: conv = nm_begin_row<1>(s, f, G.xatol, G.fatol, xr); break;
    case 2: conv = nm_begin_row<2>(s, f, G.xatol, G.fatol, xr); break;
    case 3: conv = nm_begin_row<3>(s, f, G.xatol, G.fatol, xr); break;
    case 4: conv = nm_begin_row<4>(s, f, G.xatol, G.fatol, xr); break;
    case 5: conv = nm_begin_row<5>(s, f, G.xatol, G.fatol, xr); break;
    case 6: conv = nm_begin_row<6>(s, f, G.xatol, G.fatol, xr); break;
    case 7: conv = nm_begin_row<7>(s, f, G.xatol, G.fatol, xr); break;
    default: conv = nm_begin_row<8>(s, f, G.xatol, G.fatol, xr); break;
  }
  if (conv) {
    flags[r] = (flags[r] & ~1) | 2;  // converged: success
    return 0;
  }
  return 1;
}

// nm_update_kernel's accepted point + nm_order + nm_begin_kernel's test of simplex r on
// ONE copy of the simplex in registers: the point becomes row N, the rows are ordered
// and stored, and the next round's test and reflection point come off the same
// registers -- the same operations on the same values as the three steps through
// memory (store row N; load, sort, store; load, test), two round trips shorter.
template <int N>
__device__ __forceinline__ int glue_accept_row_t(const NmGlue &G, int r, double *gs,
                                                 double *gf,
                                                 const double *__restrict__ src,
                                                 double fnew, double *xr) {
  double s[N + 1][N], f[N + 1];
#pragma unroll
  for (int a = 0; a < N; a++) {
    f[a] = gf[a];
#pragma unroll
    for (int i = 0; i < N; i++) s[a][i] = gs[a * N + i];
  }
#pragma unroll
  for (int i = 0; i < N; i++) s[N][i] = src[i];
  f[N] = fnew;
  nm_sort_regs<N>(s, f);
  nm_store_regs<N>(gs, gf, s, f);
  const int nit = G.m.nit[r] + 1;
  G.m.nit[r] = nit;
  // (glue_begin_row's tests, in its order)
  int32_t *flags = G.m.flags;
  const int fl = flags[r];
  if ((fl & 5) != 1) return 0;
  if (nit >= G.maxiter) {
    flags[r] = fl & ~1;
    return 0;
  }
  if (nm_test_regs<N>(s, f, G.xatol, G.fatol, xr)) {
    flags[r] = (fl & ~1) | 2;
    return 0;
  }
  return 1;
}

__device__ __forceinline__ int glue_accept_row(const NmGlue &G, int r, double *s,
                                               double *f, const double *src,
                                               double fnew, double *xr) {
  switch (G.m.N) {
    case 1: return glue_accept_row_t<1>(G, r, s, f, src, fnew, xr);
    case 2: return glue_accept_row_t<2>(G, r, s, f, src, fnew, xr);
    case 3: return glue_accept_row_t<3>(G, r, s, f, src, fnew, xr);
    case 4: return glue_accept_row_t<4>(G, r, s, f, src, fnew, xr);
    case 5: return glue_accept_row_t<5>(G, r, s, f, src, fnew, xr);
    case 6: return glue_accept_row_t<6>(G, r, s, f, src, fnew, xr);
    case 7: return glue_accept_row_t<7>(G, r, s, f, src, fnew, xr);
    default: return glue_accept_row_t<8>(G, r, s, f, src, fnew, xr);
  }
}

__global__ void __launch_bounds__(NM_UNT) nm_glue_begin_kernel(NmGlue G) {
  __shared__ int sh[NM_UNT / 64 + 1];
  const int S = G.m.S, N = G.m.N;
  int base_out = 0;
  for (int r0 = 0; r0 < S; r0 += NM_UNT) {
    const int r = r0 + threadIdx.x;
    double xr[NM_MAXN] = {};
    const int go = (r < S) ? glue_begin_row(G, r, xr) : 0;
    int tot;
    const int pos = base_out + block_excl_scan<NM_UNT>(go, &tot, sh);
    if (go) {
      G.m.list1[pos] = r;
      double *x = G.m.X1 + (int64_t)pos * N;
      nm_put_row(x, xr, N);
      map_row_regs(G.P, pos, r, xr);
    }
    base_out += tot;
  }
  if (threadIdx.x == 0) {
    G.m.counts[0] = base_out;
    G.m.counts[3] = base_out;
  }
}

__global__ void __launch_bounds__(NM_NT) nm_glue_decide_kernel(NmGlue G, int jbound) {
  __shared__ int sh[NM_NT / 64 + 1];
  const int N = G.m.N;
  const int J = min(G.m.counts[0], jbound);
  int base_out = 0;
  for (int j0 = 0; j0 < J; j0 += NM_NT) {
    const int j = j0 + threadIdx.x;
    int go = 0, r = 0;
    double x2[NM_MAXN] = {};
    if (j < J) {   // (nm_decide_kernel)
      r = G.m.list1[j];
      const double *s = G.m.sim + (int64_t)r * (N + 1) * N;
      const double *f = G.m.fsim + (int64_t)r * (N + 1);
      // (the simplex values requested ahead of the evaluation's chain of loads)
      const double f0 = f[0], fn1 = f[N - 1], fn = f[N];
      // (the row's own evaluation, read from job slot j: the slots this kernel
      // rewrites for the second evaluation lie at or before the rows already read)
      const double fxr = glue_value(G, j);
      G.m.F1[j] = fxr;
      int c;
      if (fxr < f0)
        c = 1;
      else if (fxr < fn1)
        c = 0;
      else if (fxr < fn)
        c = 2;
      else
        c = 3;
      G.m.cases[j] = c;
      if (c != 0) {
        go = 1;
#pragma unroll
        for (int i = 0; i < NM_MAXN; i++) {   // (x2[] in registers: static indices)
          if (i >= N) break;
          double xb = s[i];
          for (int k = 1; k < N; k++) xb = xb + s[k * N + i];
          xb = xb / N;
          const double w = s[N * N + i];
          if (c == 1)
            x2[i] = (1 + 1.0 * 2.0) * xb - 1.0 * 2.0 * w;
          else if (c == 2)
            x2[i] = (1 + 0.5 * 1.0) * xb - 0.5 * 1.0 * w;
          else
            x2[i] = (1 - 0.5) * xb + 0.5 * w;
        }
      }
    }
    int tot;
    const int pos = base_out + block_excl_scan(go, &tot, sh);
    if (j < J) G.m.pos2[j] = go ? pos : -1;
    if (go) {
      G.m.list2[pos] = r;
      nm_put_row(G.m.X2 + (int64_t)pos * N, x2, N);
      map_row_regs(G.P, pos, r, x2);
    }
    base_out += tot;
  }
  if (threadIdx.x == 0) G.m.counts[1] = base_out;
}

__global__ void __launch_bounds__(NM_UNT) nm_glue_update_kernel(NmGlue G, int jbound) {
  __shared__ int sh[NM_UNT / 64 + 1];
  __shared__ int parked;
  const int N = G.m.N;
  const int J = min(G.m.counts[0], jbound), J2 = min(G.m.counts[1], jbound);
  if (threadIdx.x == 0) parked = 0;
  // (all values of the second evaluation first: the rows' new slots below overwrite
  // the job tables glue_value reads -- a later trip's p2 can lie under an earlier
  // trip's packed positions)
  for (int p = threadIdx.x; p < J2; p += NM_UNT) G.m.F2[p] = glue_value(G, p);
  __syncthreads();
  int base_out = 0;
  for (int j0 = 0; j0 < J; j0 += NM_UNT) {
    const int j = j0 + threadIdx.x;
    int go = 0, r = 0;
    double xr[NM_MAXN] = {};
    if (j < J) {   // (nm_update_kernel)
      r = G.m.list1[j];
      double *s = G.m.sim + (int64_t)r * (N + 1) * N;
      double *f = G.m.fsim + (int64_t)r * (N + 1);
      const int c = G.m.cases[j];
      const double fxr = G.m.F1[j];
      const int p2 = G.m.pos2[j];
      const double f2 = (p2 >= 0) ? G.m.F2[p2] : __builtin_inf();
      bool take2 = false, taker = false;
      if (c == 0)
        taker = true;
      else if (c == 1) {
        if (f2 < fxr)
          take2 = true;
        else
          taker = true;
      } else if (c == 2)
        take2 = (f2 <= fxr);
      else
        take2 = (f2 < f[N]);
      G.m.nfev[r] += (c == 0) ? 1 : 2;
      if (take2 || taker) {
        const double *src = take2 ? (G.m.X2 + (int64_t)p2 * N)
                                  : (G.m.X1 + (int64_t)j * N);
        // ... and the next round's test of this simplex (nm_begin_kernel)
        go = glue_accept_row(G, r, s, f, src, take2 ? f2 : fxr, xr);
      } else {
        G.m.flags[r] |= 4;  // shrink: parked until the host runs the shrink
        atomicAdd(&parked, 1);
      }
    }
    int tot;
    const int pos = base_out + block_excl_scan<NM_UNT>(go, &tot, sh);
    if (go) {   // pos <= j: rows of this trip were read above, later rows lie behind
      G.m.list1[pos] = r;
      double *x = G.m.X1 + (int64_t)pos * N;
      nm_put_row(x, xr, N);
      map_row_regs(G.P, pos, r, xr);
    }
    base_out += tot;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    G.m.counts[0] = base_out;
    G.m.counts[3] = base_out;
    G.m.counts[4] += parked;  // simplices waiting to shrink
  }
}

// ---------------------------------------------------------------------------
// The optimiser's LAST rounds -- a few stragglers stepping, every kernel of a round a
// latency, the chip idle -- as ONE evaluation launch and ONE bookkeeping kernel per
// round: all four points a step can ask for (reflection, expansion, outside and inside
// contraction: functions of the simplex alone) are evaluated together, rows q J + j for
// candidate q of list row j, and this kernel does what decide and update do, reading
// the one second value scipy would have computed.  Same points, same values, same
// updates, nfev counted as scipy counts: the state is the two-launch round's to the
// bit; the status bits of candidates scipy would not have evaluated are not taken.
// One wave (512 VGPRs: a simplex, its next point and the three other candidates in
// registers at any N): the host sends rounds of at most NM_SNT rows here (option
// nm_spec_max).
// ---------------------------------------------------------------------------
#define NM_SNT 64
template <int N>
__device__ __forceinline__ void nm_other_points(const double (&s)[N + 1][N],
                                                double (*xc)[NM_MAXN]) {
#pragma unroll
  for (int i = 0; i < N; i++) {   // (nm_decide_kernel's expressions)
    double xb = s[0][i];
#pragma unroll
    for (int k = 1; k < N; k++) xb = xb + s[k][i];
    xb = xb / N;
    const double w = s[N][i];
    xc[0][i] = (1 + 1.0 * 2.0) * xb - 1.0 * 2.0 * w;
    xc[1][i] = (1 + 0.5 * 1.0) * xb - 0.5 * 1.0 * w;
    xc[2][i] = (1 - 0.5) * xb + 0.5 * w;
  }
}

// rows (q + 1) * J + pos of X1 and of the job tables <- candidate q of simplex r
__device__ __forceinline__ void nm_put_candidates(const NmGlue &G, int J, int pos, int r,
                                                  const double (*xc)[NM_MAXN]) {
  const int N = G.m.N;
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const int row = (q + 1) * J + pos;
    nm_put_row(G.m.X1 + (int64_t)row * N, xc[q], N);
    map_row_regs(G.P, row, r, xc[q]);
  }
}

template <int N>
__device__ __forceinline__ void glue_spec_prep_body(const NmGlue &G, int jbound) {
  const int J = min(G.m.counts[0], jbound);
  const int j = threadIdx.x;
  if (j < J) {
    const int r = G.m.list1[j];
    double s[N + 1][N], f[N + 1];
    nm_load_regs<N>(G.m.sim + (int64_t)r * (N + 1) * N, G.m.fsim + (int64_t)r * (N + 1),
                    s, f);
    double xc[3][NM_MAXN] = {};
    nm_other_points<N>(s, xc);
    nm_put_candidates(G, J, j, r, xc);
  }
  if (threadIdx.x == 0) G.m.counts[5] = 4 * J;
}

__global__ void __launch_bounds__(NM_SNT) nm_glue_spec_prep_kernel(NmGlue G, int jbound) {
  switch (G.m.N) {
    case 1: glue_spec_prep_body<1>(G, jbound); break;
    case 2: glue_spec_prep_body<2>(G, jbound); break;
    case 3: glue_spec_prep_body<3>(G, jbound); break;
    case 4: glue_spec_prep_body<4>(G, jbound); break;
    case 5: glue_spec_prep_body<5>(G, jbound); break;
    case 6: glue_spec_prep_body<6>(G, jbound); break;
    case 7: glue_spec_prep_body<7>(G, jbound); break;
    default: glue_spec_prep_body<8>(G, jbound); break;
  }
}

template <int N>
__device__ __forceinline__ void glue_spec_body(const NmGlue &G, int jbound, int *sh,
                                               int *parked) {
  const int J = min(G.m.counts[0], jbound);
  const int j = threadIdx.x;
  int go = 0, r = 0;
  double xr[NM_MAXN] = {};
  double xc[3][NM_MAXN] = {};
  if (j < J) {
    r = G.m.list1[j];
    double *gs = G.m.sim + (int64_t)r * (N + 1) * N;
    double *gf = G.m.fsim + (int64_t)r * (N + 1);
    const double f0 = gf[0], fn1 = gf[N - 1], fn = gf[N];
    const double fxr = glue_value(G, j);   // (nm_decide_kernel)
    G.m.F1[j] = fxr;
    int c;
    if (fxr < f0)
      c = 1;
    else if (fxr < fn1)
      c = 0;
    else if (fxr < fn)
      c = 2;
    else
      c = 3;
    // the step's second point: candidate c of this row, or none
    const int p2 = (c == 0) ? -1 : c * J + j;
    double f2 = __builtin_inf();
    if (p2 >= 0) f2 = glue_value(G, p2);
    bool take2 = false, taker = false;   // (nm_update_kernel)
    if (c == 0)
      taker = true;
    else if (c == 1) {
      if (f2 < fxr)
        take2 = true;
      else
        taker = true;
    } else if (c == 2)
      take2 = (f2 <= fxr);
    else
      take2 = (f2 < fn);
    G.m.nfev[r] += (c == 0) ? 1 : 2;
    if (take2 || taker) {
      const double *src = G.m.X1 + (int64_t)(take2 ? p2 : j) * N;
      double s[N + 1][N], f[N + 1];
#pragma unroll
      for (int a = 0; a < N; a++) {
        f[a] = gf[a];
#pragma unroll
        for (int i = 0; i < N; i++) s[a][i] = gs[a * N + i];
      }
#pragma unroll
      for (int i = 0; i < N; i++) s[N][i] = src[i];
      f[N] = take2 ? f2 : fxr;
      nm_sort_regs<N>(s, f);
      nm_store_regs<N>(gs, gf, s, f);
      const int nit = G.m.nit[r] + 1;
      G.m.nit[r] = nit;
      // (glue_begin_row's tests, in its order)
      int32_t *flags = G.m.flags;
      const int fl = flags[r];
      if ((fl & 5) == 1) {
        if (nit >= G.maxiter) {
          flags[r] = fl & ~1;
        } else if (nm_test_regs<N>(s, f, G.xatol, G.fatol, xr)) {
          flags[r] = (fl & ~1) | 2;
        } else {
          go = 1;
          nm_other_points<N>(s, xc);
        }
      }
    } else {
      G.m.flags[r] |= 4;  // shrink: parked until the host runs the shrink
      atomicAdd(parked, 1);
    }
  }
  // (every row and value of this round has been read: the barriers of the scan)
  int tot;
  const int pos = block_excl_scan<NM_SNT>(go, &tot, sh);
  if (go) {
    G.m.list1[pos] = r;
    nm_put_row(G.m.X1 + (int64_t)pos * N, xr, N);
    map_row_regs(G.P, pos, r, xr);
    nm_put_candidates(G, tot, pos, r, xc);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    G.m.counts[0] = tot;
    G.m.counts[3] = tot;
    G.m.counts[5] = 4 * tot;
    G.m.counts[4] += *parked;  // simplices waiting to shrink
  }
}

__global__ void __launch_bounds__(NM_SNT) nm_glue_spec_kernel(NmGlue G, int jbound) {
  __shared__ int sh[NM_SNT / 64 + 1];
  __shared__ int parked;
  if (threadIdx.x == 0) parked = 0;
  __syncthreads();
  switch (G.m.N) {
    case 1: glue_spec_body<1>(G, jbound, sh, &parked); break;
    case 2: glue_spec_body<2>(G, jbound, sh, &parked); break;
    case 3: glue_spec_body<3>(G, jbound, sh, &parked); break;
    case 4: glue_spec_body<4>(G, jbound, sh, &parked); break;
    case 5: glue_spec_body<5>(G, jbound, sh, &parked); break;
    case 6: glue_spec_body<6>(G, jbound, sh, &parked); break;
    case 7: glue_spec_body<7>(G, jbound, sh, &parked); break;
    default: glue_spec_body<8>(G, jbound, sh, &parked); break;
  }
}

// ---------------------------------------------------------------------------
// The two bookkeeping kernels above as TWO kernels each, for rounds of thousands of
// rows: everything a row does for itself -- the value of its evaluation, its case,
// the simplex update and ordering, the next round's test and point -- on as many
// blocks as there are rows (`rows`), and the ordered compaction (lists in simplex
// order, in place, as the one-block kernels leave them) behind it on one block that
// only moves rows (`pack`).  At 5000 rows the one-block kernels take 160 and 75 us --
// a fifth of a half's timeline, hidden only while the other half's objective kernel
// runs.  Same operations per row, same order of the lists: the same state.  A row's
// point waits in ITS OWN row of X1 / X2 (no other row reads that one) and its flag in
// cases[]; the pack kernel reads a trip's rows into registers, waits for them, and
// only then writes the packed positions (which lie at or before the rows read).
// ---------------------------------------------------------------------------
#define NM_ROWS_NT 256
__global__ void __launch_bounds__(NM_ROWS_NT)
    nm_glue_decide_rows_kernel(NmGlue G, int jbound) {
  const int N = G.m.N;
  const int J = min(G.m.counts[0], jbound);
  const int j = blockIdx.x * NM_ROWS_NT + threadIdx.x;
  if (j >= J) return;
  const int r = G.m.list1[j];
  const double *s = G.m.sim + (int64_t)r * (N + 1) * N;
  const double *f = G.m.fsim + (int64_t)r * (N + 1);
  // (the simplex values requested ahead of the evaluation's chain of loads)
  const double f0 = f[0], fn1 = f[N - 1], fn = f[N];
  const double fxr = glue_value(G, j);
  G.m.F1[j] = fxr;
  int c;
  if (fxr < f0)
    c = 1;
  else if (fxr < fn1)
    c = 0;
  else if (fxr < fn)
    c = 2;
  else
    c = 3;
  G.m.cases[j] = c;
  if (c != 0) {
    double *x2 = G.m.X2 + (int64_t)j * N;
    for (int i = 0; i < N; i++) {
      double xb = s[i];
      for (int k = 1; k < N; k++) xb = xb + s[k * N + i];
      xb = xb / N;
      const double w = s[N * N + i];
      if (c == 1)
        x2[i] = (1 + 1.0 * 2.0) * xb - 1.0 * 2.0 * w;
      else if (c == 2)
        x2[i] = (1 + 0.5 * 1.0) * xb - 0.5 * 1.0 * w;
      else
        x2[i] = (1 - 0.5) * xb + 0.5 * w;
    }
  }
}

// rows [0, J) with flag(j) set move, in order, to the front of (list, X): X row j -> row
// pos, list[pos] = list_in[j] (j itself without a list); pos_out[j] = pos or -1
// (nullable); returns the count
template <typename FLAG>
__device__ int nm_pack_rows(const NmGlue &G, int J, FLAG flag, const int32_t *list_in,
                            int32_t *list_out, double *X, int32_t *pos_out, int *sh) {
  const int N = G.m.N;
  int base_out = 0;
  for (int j0 = 0; j0 < J; j0 += NM_NT) {
    const int j = j0 + threadIdx.x;
    int go = 0, r = 0;
    double xr[NM_MAXN] = {};
    if (j < J) {
      go = flag(j);
      r = list_in ? list_in[j] : j;
      if (go) nm_get_row(xr, X + (int64_t)j * N, N);
    }
    // (the rows are in registers before any thread writes a packed position)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int tot;
    const int pos = base_out + block_excl_scan(go, &tot, sh);
    if (pos_out && j < J) pos_out[j] = go ? pos : -1;
    if (go) {
      list_out[pos] = r;
      nm_put_row(X + (int64_t)pos * N, xr, N);
      map_row_regs(G.P, pos, r, xr);
    }
    base_out += tot;
  }
  return base_out;
}

// nm_glue_begin_kernel the same way (the start of a run and every return from a shrink
// test all S simplices: ten trips of one block at 5000)
__global__ void __launch_bounds__(NM_ROWS_NT) nm_glue_begin_rows_kernel(NmGlue G) {
  const int S = G.m.S, N = G.m.N;
  const int r = blockIdx.x * NM_ROWS_NT + threadIdx.x;
  if (r >= S) return;
  double xr[NM_MAXN] = {};
  const int go = glue_begin_row(G, r, xr);
  G.m.cases[r] = go;   // (no round is in flight: the array is free)
  if (go) nm_put_row(G.m.X1 + (int64_t)r * N, xr, N);
}

__global__ void __launch_bounds__(NM_NT) nm_glue_begin_pack_kernel(NmGlue G) {
  __shared__ int sh[NM_NT / 64 + 1];
  const int32_t *cases = G.m.cases;
  const int n = nm_pack_rows(G, G.m.S, [=](int j) { return cases[j]; }, nullptr,
                             G.m.list1, G.m.X1, nullptr, sh);
  __syncthreads();
  if (threadIdx.x == 0) {
    G.m.counts[0] = n;
    G.m.counts[3] = n;
  }
}

__global__ void __launch_bounds__(NM_NT) nm_glue_decide_pack_kernel(NmGlue G, int jbound) {
  __shared__ int sh[NM_NT / 64 + 1];
  const int J = min(G.m.counts[0], jbound);
  const int32_t *cases = G.m.cases;
  const int n = nm_pack_rows(G, J, [=](int j) { return cases[j] != 0 ? 1 : 0; },
                             G.m.list1, G.m.list2, G.m.X2, G.m.pos2, sh);
  if (threadIdx.x == 0) G.m.counts[1] = n;
}

__global__ void __launch_bounds__(NM_ROWS_NT)
    nm_glue_update_rows_kernel(NmGlue G, int jbound) {
  const int N = G.m.N;
  const int J = min(G.m.counts[0], jbound);
  const int j = blockIdx.x * NM_ROWS_NT + threadIdx.x;
  if (j >= J) return;
  const int r = G.m.list1[j];
  double *s = G.m.sim + (int64_t)r * (N + 1) * N;
  double *f = G.m.fsim + (int64_t)r * (N + 1);
  const int c = G.m.cases[j];
  const double fxr = G.m.F1[j];
  const int p2 = G.m.pos2[j];
  const double fn = f[N];   // (requested ahead of the evaluation's chain of loads)
  double f2 = __builtin_inf();
  if (p2 >= 0) {   // (this row's second point: nobody else's)
    f2 = glue_value(G, p2);
    G.m.F2[p2] = f2;
  }
  bool take2 = false, taker = false;
  if (c == 0)
    taker = true;
  else if (c == 1) {
    if (f2 < fxr)
      take2 = true;
    else
      taker = true;
  } else if (c == 2)
    take2 = (f2 <= fxr);
  else
    take2 = (f2 < fn);
  G.m.nfev[r] += (c == 0) ? 1 : 2;
  int go = 0;
  double xr[NM_MAXN];
  if (take2 || taker) {
    const double *src = take2 ? (G.m.X2 + (int64_t)p2 * N) : (G.m.X1 + (int64_t)j * N);
    go = glue_accept_row(G, r, s, f, src, take2 ? f2 : fxr, xr);
  } else {
    G.m.flags[r] |= 4;  // shrink: parked until the host runs the shrink
    atomicAdd(&G.m.counts[4], 1);
  }
  G.m.cases[j] = go;   // (the case is used up: the flag of the pack kernel)
  if (go) nm_put_row(G.m.X1 + (int64_t)j * N, xr, N);
}

__global__ void __launch_bounds__(NM_NT) nm_glue_update_pack_kernel(NmGlue G, int jbound) {
  __shared__ int sh[NM_NT / 64 + 1];
  const int J = min(G.m.counts[0], jbound);
  const int32_t *cases = G.m.cases;
  const int n = nm_pack_rows(G, J, [=](int j) { return cases[j]; }, G.m.list1,
                             G.m.list1, G.m.X1, nullptr, sh);
  __syncthreads();
  if (threadIdx.x == 0) {
    G.m.counts[0] = n;
    G.m.counts[3] = n;
  }
}

static int nm_bucket(int n, int S) {
  // quantised launch bound (1/8 steps of the next power of two), as optimizer.py
  if (n <= 64) return S < 64 ? S : 64;
  int p2 = 1;
  while (p2 < n) p2 <<= 1;
  const int stepq = (p2 / 8 > 1) ? p2 / 8 : 1;
  const int b = ((n + stepq - 1) / stepq) * stepq;
  return b < S ? b : S;
}

// the evaluation of the rows the last bookkeeping kernel mapped (J = launch bound,
// `live` = their count on the device); the per-arm results stay in o->scratch
static int nm_objective_rows(const rvs_nm_objective *o, int J, const int32_t *live,
                             hipStream_t st) {
  if (o->nn) {
    const double *tp[8], *op[8];
    if (o->narm > 8) return RVS_E_ARG;
    int rc = rvs_template_nn_arms_n(o->params, J, live, o->ndim, o->narm, o->nn, st);
    if (rc) return rc;
    for (int a = 0; a < o->narm; a++) {
      tp[a] = o->nn[a].templ;
      op[a] = o->nn[a].outside;
    }
    return rvs_objective_from_template_n(o->arms, o->narm, o->npoly, tp, op,
                                         o->vsini, o->job_spec, J, live, o->vel,
                                         o->badchi, 1 | RVS_OBJ_NO_SUM, o->scratch,
                                         nullptr, nullptr, st);
  }
  if (o->tri) {
    const double *tp[8], *op[8];
    if (o->narm > 8) return RVS_E_ARG;
    int rc = rvs_internal_template_tri_arms_n(o->params, J, live, o->ndim, o->narm,
                                              o->tri, st);
    if (rc) return rc;
    for (int a = 0; a < o->narm; a++) {
      tp[a] = o->tri[a].templ;
      op[a] = o->tri[a].outside;
    }
    return rvs_objective_from_template_n(o->arms, o->narm, o->npoly, tp, op,
                                         o->vsini, o->job_spec, J, live, o->vel,
                                         o->badchi, 1 | RVS_OBJ_NO_SUM, o->scratch,
                                         nullptr, nullptr, st);
  }
  return rvs_objective_fused_n(o->arms, o->narm, o->npoly, o->params, o->vsini,
                               o->job_spec, J, live, o->vel, o->badchi,
                               1 | RVS_OBJ_NO_SUM, o->scratch, nullptr, nullptr, st);
}

static int nm_run_chain(const rvs_nm_state *m, const rvs_nm_objective *o,
                        double xatol, double fatol, int maxiter, int sync_every,
                        int64_t *stats, void *stream);

extern "C" int rvs_nm_run(const rvs_nm_state *m, const rvs_nm_objective *o,
                          double xatol, double fatol, int maxiter,
                          int sync_every, int64_t *stats, void *stream) {
  if (!m || !o || m->S < 1 || m->N < 1 || m->N > NM_MAXN || sync_every < 1)
    return RVS_E_ARG;
  // nm_glue = 0: the round as a chain of the stand-alone kernels (a test hook:
  // tests/test_gpu_parity.py::test_nm_round_kernels_equal_chain)
  if (!rvs_opt(RVS_OPT_NM_GLUE))
    return nm_run_chain(m, o, xatol, fatol, maxiter, sync_every, stats, stream);
  hipStream_t st = rvs_stream(stream);
  const int S = m->S, N = m->N;
  int64_t rounds = 0, calls = 0, jobs = 0;
  int32_t c[8];
  NmGlue G;
  G.m = *m;
  MapP &P = G.P;
  P.n = o->n, P.ndim = o->ndim, P.vsini_col = o->vsini_col;
  for (int i = 0; i < NM_MAXN; i++) P.M.src[i] = (i < o->ndim) ? o->src[i] : -1;
  P.fixed = o->fixed, P.vsini_fixed = o->vsini_fixed, P.safe = o->safe;
  P.prior_mean = o->prior_mean, P.prior_isig = o->prior_isig;
  P.min_vel = o->min_vel, P.max_vel = o->max_vel, P.max_vsini = o->max_vsini;
  P.job_spec = o->job_spec, P.vel = o->vel, P.vsini = o->vsini, P.params = o->params;
  P.extra = o->extra, P.bad = o->bad;
  G.badchi = o->badchi, G.xatol = xatol, G.fatol = fatol, G.maxiter = maxiter;
  G.pen_scale = o->arms[0].pt.pen_scale;
  G.spec_status = o->status;
  G.AO = obj_arm_out(o->scratch, o->narm, S);
  auto begin = [&]() {   // every simplex tested, the live ones listed and mapped
    if (S >= rvs_opt(RVS_OPT_NM_SPLIT_MIN)) {
      hipLaunchKernelGGL(nm_glue_begin_rows_kernel,
                         dim3((S + NM_ROWS_NT - 1) / NM_ROWS_NT), dim3(NM_ROWS_NT), 0, st,
                         G);
      hipLaunchKernelGGL(nm_glue_begin_pack_kernel, dim3(1), dim3(NM_NT), 0, st, G);
    } else {
      hipLaunchKernelGGL(nm_glue_begin_kernel, dim3(1), dim3(NM_UNT), 0, st, G);
    }
  };
  begin();
  RVS_LAUNCH_CHECK();
  int rc = 0;
  while (true) {
    if (hipMemcpyAsync(c, m->counts, sizeof(c), hipMemcpyDeviceToHost, st) !=
            hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return RVS_E_LAUNCH;
    const int live = c[0], parked = c[4];
    if (parked > 0) {  // scipy's shrink step for the parked simplices
      rc = rvs_nm_collect(S, m->flags, m->list3, m->counts, st);
      if (rc) return rc;
      for (int k = 1; k <= N; k++) {
        rc = rvs_nm_shrink_point(N, k, m->sim, m->list3, m->X2, m->counts,
                                 parked, st);
        if (rc) return rc;
        rc = rvs_internal_nm_eval(o, m->list3, m->X2, parked, m->counts, 2, m->F2, st);
        if (rc) return rc;
        calls++;
        jobs += parked;
        rc = rvs_nm_shrink_store(N, k, m->sim, m->fsim, m->nit, m->nfev,
                                 m->flags, m->list3, m->F2, m->counts, parked,
                                 st);
        if (rc) return rc;
      }
      begin();
      RVS_LAUNCH_CHECK();
      continue;
    }
    if (live == 0) break;
    // (the caller takes the finished spectra on and calls again for the rest)
    if (m->stop_below > 0 && live <= m->stop_below) break;
    // the live count only falls between two looks (finished and parked
    // simplices leave the list), so it bounds the launches of the window
    const int jb = live;
    // (the host looks -- a copy of the counters and a stream synchronisation, then the
    // next launches: ~25 us with nothing queued -- every sync_every rounds while the
    // launch bound matters, and less often in the latency-bound last rounds, where a
    // block behind the live count costs nothing.  A simplex's path does not depend on
    // when the host looks)
    const int window = (jb <= 256) ? max(sync_every, rvs_opt(RVS_OPT_NM_TAIL_WINDOW))
                                   : sync_every;
    // a handful of stragglers: the step's four candidate points in one launch and one
    // bookkeeping kernel per round (nm_glue_spec_kernel)
    if (jb <= rvs_opt(RVS_OPT_NM_SPEC_MAX) && jb <= NM_SNT && 4 * (int64_t)jb <= S) {
      G.AO = obj_arm_out(o->scratch, o->narm, 4 * jb);
      hipLaunchKernelGGL(nm_glue_spec_prep_kernel, dim3(1), dim3(NM_SNT), 0, st, G, jb);
      RVS_LAUNCH_CHECK();
      for (int r = 0; r < window; r++) {
        rc = nm_objective_rows(o, 4 * jb, m->counts + 5, st);
        if (rc) return rc;
        hipLaunchKernelGGL(nm_glue_spec_kernel, dim3(1), dim3(NM_SNT), 0, st, G, jb);
        RVS_LAUNCH_CHECK();
        calls += 1;
        jobs += 4 * (int64_t)jb;
      }
      rounds += window;
      continue;
    }
    G.AO = obj_arm_out(o->scratch, o->narm, jb);
    for (int r = 0; r < window; r++) {
      // (from a few thousand rows up the bookkeeping is row-parallel + pack)
      const bool split = jb >= rvs_opt(RVS_OPT_NM_SPLIT_MIN);
      const dim3 rgrid((jb + NM_ROWS_NT - 1) / NM_ROWS_NT);
      rc = nm_objective_rows(o, jb, m->counts, st);
      if (rc) return rc;
      if (split) {
        hipLaunchKernelGGL(nm_glue_decide_rows_kernel, rgrid, dim3(NM_ROWS_NT), 0, st,
                           G, jb);
        hipLaunchKernelGGL(nm_glue_decide_pack_kernel, dim3(1), dim3(NM_NT), 0, st, G,
                           jb);
      } else {
        hipLaunchKernelGGL(nm_glue_decide_kernel, dim3(1), dim3(NM_NT), 0, st, G, jb);
      }
      RVS_LAUNCH_CHECK();
      rc = nm_objective_rows(o, jb, m->counts + 1, st);
      if (rc) return rc;
      if (split) {
        hipLaunchKernelGGL(nm_glue_update_rows_kernel, rgrid, dim3(NM_ROWS_NT), 0, st,
                           G, jb);
        hipLaunchKernelGGL(nm_glue_update_pack_kernel, dim3(1), dim3(NM_NT), 0, st, G,
                           jb);
      } else {
        hipLaunchKernelGGL(nm_glue_update_kernel, dim3(1), dim3(NM_UNT), 0, st, G, jb);
      }
      RVS_LAUNCH_CHECK();
      calls += 2;
      jobs += 2 * (int64_t)jb;
    }
    rounds += window;
  }
  if (stats) {
    stats[0] = rounds;
    stats[1] = calls;
    stats[2] = jobs;
  }
  return 0;
}

static int nm_run_chain(const rvs_nm_state *m, const rvs_nm_objective *o,
                        double xatol, double fatol, int maxiter, int sync_every,
                        int64_t *stats, void *stream) {
  hipStream_t st = rvs_stream(stream);
  const int S = m->S, N = m->N;
  int64_t rounds = 0, calls = 0, jobs = 0;
  int32_t c[8];
  int rc = rvs_nm_begin(S, N, xatol, fatol, maxiter, m->sim, m->fsim, m->nit,
                        m->flags, m->list1, m->X1, m->counts, S, st);
  if (rc) return rc;
  while (true) {
    if (hipMemcpyAsync(c, m->counts, sizeof(c), hipMemcpyDeviceToHost, st) !=
            hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return RVS_E_LAUNCH;
    const int live = c[0], parked = c[4];
    if (parked > 0) {  // scipy's shrink step for the parked simplices
      rc = rvs_nm_collect(S, m->flags, m->list3, m->counts, st);
      if (rc) return rc;
      for (int k = 1; k <= N; k++) {
        rc = rvs_nm_shrink_point(N, k, m->sim, m->list3, m->X2, m->counts,
                                 parked, st);
        if (rc) return rc;
        rc = rvs_internal_nm_eval(o, m->list3, m->X2, parked, m->counts, 2, m->F2, st);
        if (rc) return rc;
        calls++;
        jobs += parked;
        rc = rvs_nm_shrink_store(N, k, m->sim, m->fsim, m->nit, m->nfev,
                                 m->flags, m->list3, m->F2, m->counts, parked,
                                 st);
        if (rc) return rc;
      }
      rc = rvs_nm_begin(S, N, xatol, fatol, maxiter, m->sim, m->fsim, m->nit,
                        m->flags, m->list1, m->X1, m->counts, S, st);
      if (rc) return rc;
      continue;
    }
    if (live == 0) break;
    // the live count only falls between two looks (finished and parked
    // simplices leave the list), so it bounds the launches of the window
    const int jb = rvs_opt(RVS_OPT_NM_BUCKET) ? nm_bucket(live, S) : live;
    for (int r = 0; r < sync_every; r++) {
      rc = rvs_nm_begin(S, N, xatol, fatol, maxiter, m->sim, m->fsim, m->nit,
                        m->flags, m->list1, m->X1, m->counts, jb, st);
      if (rc) return rc;
      rc = rvs_internal_nm_eval(o, m->list1, m->X1, jb, m->counts, 0, m->F1, st);
      if (rc) return rc;
      rc = rvs_nm_decide(N, m->sim, m->fsim, m->list1, m->F1, m->cases,
                         m->pos2, m->list2, m->X2, m->counts, jb, st);
      if (rc) return rc;
      rc = rvs_internal_nm_eval(o, m->list2, m->X2, jb, m->counts, 1, m->F2, st);
      if (rc) return rc;
      rc = rvs_nm_update(N, m->sim, m->fsim, m->nit, m->nfev, m->list1, m->X1,
                         m->F1, m->cases, m->pos2, m->X2, m->F2, m->flags,
                         m->counts, jb, st);
      if (rc) return rc;
      calls += 2;
      jobs += 2 * (int64_t)jb;
    }
    rounds += sync_every;
  }
  if (stats) {
    stats[0] = rounds;
    stats[1] = calls;
    stats[2] = jobs;
  }
  return 0;
}

