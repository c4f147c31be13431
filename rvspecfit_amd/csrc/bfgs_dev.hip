// bfgs_dev.hip -- the second minimiser of vel_fit.process on the device
// (vel_fit.py:653-658: scipy.optimize.minimize(method='BFGS', hess_inv0=...) from
// the simplex optimum; utils.py:26 makes it the reference's default).
//
// S runs of bfgs_machine.h -- the state machine the CPU suite pins to scipy through
// bfgs_host.cpp -- live in HBM, one thread per spectrum advances its run to the next
// request, and a round is
//   bfgs_advance_kernel  values of the last request -> run -> rows it needs next
//   bfgs_scan_kernel     exclusive scan of the row counts: every run's first row,
//                        the rows of each launch chunk, the number of live runs
//   bfgs_emit_kernel     the requested points into one list (spectrum order)
//   <objective>          rvs_proc_map -> objective kernel -> rvs_proc_finish per chunk
//                        of `cap` rows (nm.hip: the objective of rvs_nm_run)
// with the counts on the device: the host looks (a 128-byte copy) every round while
// the launches are large, every few rounds in the tail, and only to bound the
// launches and to see the end -- a run's path does not depend on when it looks.
// The host-driven form of the same rounds (bfgs_host.cpp behind a Python objective)
// cost 1.6 ms per round in torch calls and copies: 0.35 s per 2000 spectra on 0.19 s
// of objective kernels.
#include "bfgs_machine.h"
#include "common.h"
#include "nm_internal.h"

using rvs_bfgs::Run;

#ifndef BF_NT
// advance: FOUR runs per block -- a run's state is 7.7 KB of its own, so every load of
// a wave touches as many cache lines as it has live lanes; few lanes per wave and many
// waves over the CUs: BFGS of 500 / 2000 spectra 0.088 / 0.227 -> 0.083 / 0.221 s
// against 64 runs per block (16: 0.085 / 0.224; tools/perf/ab_libs3.sh)
#define BF_NT 4
#endif
#define BF_SCAN_NT 1024
#define BF_NCHUNK 24    // counts[0 .. 24): rows of chunk c; [24] rows; [25] live runs

namespace {

struct BfgsDev {
  Run *runs;
  int S, n, cap;
  const double *x0, *H0;
  double gtol, c1, c2, xrtol;
  int maxiter;
  int32_t *nreq, *off, *list, *counts;
  double *X, *F;
};

__global__ void __launch_bounds__(BF_NT) bfgs_advance_kernel(BfgsDev D, int first) {
  const int s = blockIdx.x * BF_NT + threadIdx.x;
  if (s >= D.S) return;
  Run &r = D.runs[s];
  if (first) {
    rvs_bfgs::init(r, D.n, D.x0 + (int64_t)s * D.n, D.H0, D.gtol, D.c1, D.c2, D.xrtol,
                   D.maxiter);
  } else {
    if (r.done) return;   // (nreq[s] is 0 since the round it finished in)
    const int m = r.nrows, o = D.off[s];
    for (int q = 0; q < m; q++) r.vals[q] = D.F[o + q];
  }
  rvs_bfgs::advance(r);
  D.nreq[s] = r.done ? 0 : r.nrows;
}

__global__ void __launch_bounds__(BF_SCAN_NT) bfgs_scan_kernel(BfgsDev D) {
  __shared__ int wsum[BF_SCAN_NT / 64];
  __shared__ int carry[2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry[0] = carry[1] = 0;
  __syncthreads();
  for (int s0 = 0; s0 < D.S; s0 += BF_SCAN_NT) {
    const int s = s0 + tid;
    const int nr = (s < D.S) ? D.nreq[s] : 0;
    int v = nr;   // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(v, o, 64);
      if (lane >= o) v += t;
    }
    if (lane == 63) wsum[w] = v;
    const int alive = wave_sum_i(nr > 0 ? 1 : 0);
    __syncthreads();
    int base = carry[0];
    for (int i = 0; i < w; i++) base += wsum[i];
    if (s < D.S) D.off[s] = base + v - nr;
    __syncthreads();
    if (lane == 0 && alive) atomicAdd(&carry[1], alive);
    if (tid == BF_SCAN_NT - 1) carry[0] = base + v;
    __syncthreads();
  }
  if (tid < BF_NCHUNK) {
    const int rest = carry[0] - tid * D.cap;
    D.counts[tid] = rest < 0 ? 0 : (rest > D.cap ? D.cap : rest);
  }
  if (tid == 0) {
    D.counts[BF_NCHUNK] = carry[0];
    D.counts[BF_NCHUNK + 1] = carry[1];
  }
}

__global__ void __launch_bounds__(256) bfgs_emit_kernel(BfgsDev D) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int per = D.n + 1;
  const int s = t / per, q = t - s * per;
  if (s >= D.S || q >= D.nreq[s]) return;
  const int j = D.off[s] + q, n = D.n;
  D.list[j] = s;
  const double *src = D.runs[s].rows + q * n;
  for (int i = 0; i < n; i++) D.X[(int64_t)j * n + i] = src[i];
}

__global__ void __launch_bounds__(BF_NT)
    bfgs_result_kernel(BfgsDev D, double *x, double *fun, double *hess_inv,
                       int32_t *nit, int32_t *nfev, int32_t *status) {
  const int s = blockIdx.x * BF_NT + threadIdx.x;
  if (s >= D.S) return;
  const Run &r = D.runs[s];
  const int n = D.n;
  for (int i = 0; i < n; i++) x[(int64_t)s * n + i] = r.xk[i];
  fun[s] = r.fval;
  nit[s] = r.nit;
  nfev[s] = r.nfev;
  status[s] = r.done ? r.status : -1;
  if (hess_inv)
    for (int i = 0; i < n * n; i++) hess_inv[(int64_t)s * n * n + i] = r.Hk[i];
}

}  // namespace

extern "C" int64_t rvs_bfgs_run_bytes(void) { return (int64_t)sizeof(Run); }

extern "C" int rvs_bfgs_run(const rvs_bfgs_state *b, const rvs_nm_objective *o,
                            int sync_every, int64_t *stats, void *stream) {
  if (!b || !o || b->S < 1 || b->n < 1 || b->n > 8 || b->n != o->n || b->cap < 1 ||
      sync_every < 1 || !b->runs || !b->x0 || !b->x || !b->fun || !b->nit ||
      !b->nfev || !b->status || !b->nreq || !b->off || !b->list || !b->X || !b->F ||
      !b->counts)
    return RVS_E_ARG;
  const int S = b->S, n = b->n, cap = b->cap;
  const int64_t maxrows = (int64_t)S * (n + 1);
  if ((maxrows + cap - 1) / cap > BF_NCHUNK) return RVS_E_ARG;
  hipStream_t st = rvs_stream(stream);
  BfgsDev D;
  D.runs = static_cast<Run *>(b->runs);
  D.S = S, D.n = n, D.cap = cap;
  D.x0 = b->x0, D.H0 = b->hess_inv0;
  D.gtol = b->gtol, D.c1 = b->c1, D.c2 = b->c2, D.xrtol = b->xrtol;
  D.maxiter = b->maxiter;
  D.nreq = b->nreq, D.off = b->off, D.list = b->list, D.counts = b->counts;
  D.X = b->X, D.F = b->F;
  const dim3 agrid((S + BF_NT - 1) / BF_NT);
  const dim3 egrid((int)((maxrows + 255) / 256));
  auto step = [&](int first) {
    hipLaunchKernelGGL(bfgs_advance_kernel, agrid, dim3(BF_NT), 0, st, D, first);
    hipLaunchKernelGGL(bfgs_scan_kernel, dim3(1), dim3(BF_SCAN_NT), 0, st, D);
    hipLaunchKernelGGL(bfgs_emit_kernel, egrid, dim3(256), 0, st, D);
  };
  step(1);
  RVS_LAUNCH_CHECK();
  int64_t rounds = 0, calls = 0, jobs = 0;
  int32_t c[BF_NCHUNK + 8];
  while (true) {
    if (hipMemcpyAsync(c, b->counts, sizeof(int32_t) * (BF_NCHUNK + 2),
                       hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return RVS_E_LAUNCH;
    const int64_t total = c[BF_NCHUNK], live = c[BF_NCHUNK + 1];
    if (total == 0) break;
    // (large launches: one look per round, so that every chunk launched has rows;
    // the tail: every sync_every rounds behind the bound "every live run asks for a
    // value and a gradient", the rows behind the device count cost next to nothing)
    const int window = (total > 4096) ? 1 : sync_every;
    for (int r = 0; r < window; r++) {
      int64_t bound = (r == 0) ? total : live * (n + 1);
      if (bound > maxrows) bound = maxrows;
      for (int64_t a = 0, ch = 0; a < bound; a += cap, ch++) {
        const int J = (int)((bound - a < cap) ? bound - a : cap);
        int rc = rvs_internal_nm_eval(o, b->list + a, b->X + a * n, J, b->counts,
                                      (int)ch, b->F + a, st);
        if (rc) return rc;
        calls++;
        jobs += J;
      }
      step(0);
      RVS_LAUNCH_CHECK();
      rounds++;
    }
  }
  hipLaunchKernelGGL(bfgs_result_kernel, agrid, dim3(BF_NT), 0, st, D, b->x, b->fun,
                     b->hess_inv, b->nit, b->nfev, b->status);
  RVS_LAUNCH_CHECK();
  if (hipStreamSynchronize(st) != hipSuccess) return RVS_E_LAUNCH;
  if (stats) {
    stats[0] = rounds;
    stats[1] = calls;
    stats[2] = jobs;
  }
  return 0;
}
