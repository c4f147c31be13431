// tables.hip -- the per-wavelength-grid tables of an arm, built on the device.
//
// Everything the kernels read that depends only on an arm's wavelength grid --
// the continuum basis (spec_fit.py:148-176), its orthonormal form for the
// velocity-grid kernel, the rebin tables onto the log-lambda FFT grid
// (make_ccf.py:355-357, 394-399), the B-spline tables of the CCF continuum fit
// (make_ccf.py:155-164) and the pixel ranges of its binned-median start
// (make_ccf.py:128-143) -- was built on the host with numpy, once per arm.  With a
// grid set (every SDSS-style object on its own grid) that is once per SPECTRUM: 8 ms
// of LAPACK QR + 1 ms of searches each, 90 s for a batch the GPU fits in 0.07 s.
// Here one block per grid does the same work: binary searches with numpy's
// searchsorted semantics, the B-spline recursion of FITPACK's fpbspl in closed
// form, the basis functions, and modified Gram-Schmidt (twice) instead of
// Householder QR -- the orthonormal basis spans the same space, which is all the
// marginalised likelihood depends on.  No fused multiply-adds where
// the host rounded twice (fp contract off): apart from `exp` (one ulp) and the
// orthonormalisation the tables are the host's bit for bit.
#include "common.h"

#define TB_NT 256
#define TB_NW (TB_NT / 64)

__device__ __forceinline__ double tb_block_sum(double v, double *red) {
  // fixed order: lane butterflies, then the four waves in order
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double t = 0;
#pragma unroll
  for (int i = 0; i < TB_NW; i++) t += red[i];
  return t;
}

// np.searchsorted(a[0:n], v, side): first index with a[i] >= v ('left') or > v
__device__ __forceinline__ int tb_search(const double *a, int n, double v,
                                         bool right) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const bool go = right ? (a[mid] <= v) : (a[mid] < v);
    if (go)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// ---------------------------------------------------------------------------
// continuum basis get_basis(lam) (spec_fit.py:148-176) of every grid, pixel-major
// with one zero row behind the last pixel (and zero rows on the padding of a short
// grid), and its orthonormal form Q (P^T = Q R over the grid's own pixels) with
// 2 sum log |R_jj|.
// ---------------------------------------------------------------------------
#define TB_MAXP 32  // = FULL_MAXP of chisq.hip: the widest basis any kernel takes
// TB_PPT pixels per thread live in registers across a column's projections:
// 32 (npix <= 8192: every survey arm) or 64 (npix <= 16384)
template <int TB_PPT>
__global__ void __launch_bounds__(TB_NT)
    basis_build_kernel(const double *__restrict__ lam,
                       const int32_t *__restrict__ npix_g, int npix_s, int P,
                       int rbf, const double *__restrict__ cen,
                       double *__restrict__ raw, double *__restrict__ ortho,
                       double *__restrict__ logdet) {
#pragma clang fp contract(off)
  __shared__ double red[TB_NW];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int n = npix_g ? npix_g[g] : npix_s;
  const double *lg = lam + (int64_t)g * npix_s;
  double *R = raw + (int64_t)g * (npix_s + 1) * P;
  const double l0 = lg[0], l1 = lg[n - 1];
  const int nrbf = P - 3;
  const double sig = (nrbf > 0) ? 1. / nrbf : 1.0;
  const double sig2 = sig * sig;
  for (int k = tid; k <= npix_s; k += TB_NT) {
    double *row = R + (int64_t)k * P;
    if (k >= n) {
      for (int i = 0; i < P; i++) row[i] = 0.0;
      continue;
    }
    const double x = (lg[k] - l0) / (l1 - l0) * 2 - 1;
    if (rbf) {
      for (int i = 0; i < P && i < 3; i++) row[i] = (i == 0) ? 1.0 : (i == 1 ? x : x * x);
      for (int i = 3; i < P; i++) {
        const double a = x - cen[i - 3];
        row[i] = exp(-0.5 * (a * a) / sig2);
      }
    } else {
      // numpy.polynomial.chebyshev.chebval of the unit coefficient vector e_i
      // (Clenshaw recursion, the operations of numpy's loop in its order)
      for (int i = 0; i < P; i++) {
        double val;
        if (P == 1) {
          val = 1.0 + 0.0 * x;
        } else if (P == 2) {
          const double c0 = (i == 0), c1 = (i == 1);
          val = c0 + c1 * x;
        } else {
          const double x2 = 2 * x;
          double c0 = (i == P - 2), c1 = (i == P - 1);
          for (int ii = 3; ii <= P; ii++) {
            const double tmp = c0;
            c0 = (double)(i == P - ii) - c1;
            c1 = tmp + c1 * x2;
          }
          val = c0 + c1 * x;
        }
        row[i] = val;
      }
    }
  }
  if (!ortho) return;
  __syncthreads();
  // modified Gram-Schmidt with one re-orthogonalisation: column j against the
  // finished columns, twice; a thread keeps its pixels of the column in registers
  double *Q = ortho + (int64_t)g * (npix_s + 1) * P;
  for (int k = tid; k <= npix_s; k += TB_NT)
    if (k >= n)
      for (int i = 0; i < P; i++) Q[(int64_t)k * P + i] = 0.0;
  double ld = 0;
  for (int j = 0; j < P; j++) {
    double v[TB_PPT];
#pragma unroll
    for (int m = 0; m < TB_PPT; m++) {
      const int k = tid + m * TB_NT;
      v[m] = (k < n) ? R[(int64_t)k * P + j] : 0.0;
    }
    for (int pass = 0; pass < 2; pass++)
      for (int i = 0; i < j; i++) {
        double s = 0;
#pragma unroll
        for (int m = 0; m < TB_PPT; m++) {
          const int k = tid + m * TB_NT;
          if (k < n) s += Q[(int64_t)k * P + i] * v[m];
        }
        const double r = tb_block_sum(s, red);
#pragma unroll
        for (int m = 0; m < TB_PPT; m++) {
          const int k = tid + m * TB_NT;
          if (k < n) v[m] -= r * Q[(int64_t)k * P + i];
        }
      }
    double s2 = 0;
#pragma unroll
    for (int m = 0; m < TB_PPT; m++) s2 += v[m] * v[m];
    const double nrm = sqrt(tb_block_sum(s2, red));
    ld += log(nrm);
#pragma unroll
    for (int m = 0; m < TB_PPT; m++) {
      const int k = tid + m * TB_NT;
      if (k < n) Q[(int64_t)k * P + j] = v[m] / nrm;
    }
    __syncthreads();   // column j is complete before column j+1 projects on it
  }
  if (tid == 0) logdet[g] = 2.0 * ld;
}

extern "C" int rvs_basis_build(const double *lam, const int32_t *npix_g, int G,
                               int npix, int npoly, int rbf, const double *cen,
                               double *raw, double *ortho, double *logdet,
                               void *stream) {
  if (G < 1 || npix < 2 || npix > TB_NT * 64 || npoly < 1 || npoly > TB_MAXP ||
      !lam || !raw || (ortho && !logdet) || (rbf && npoly > 3 && !cen))
    return RVS_E_ARG;
  if (npix <= TB_NT * 32)
    hipLaunchKernelGGL(basis_build_kernel<32>, dim3(G), dim3(TB_NT), 0,
                       rvs_stream(stream), lam, npix_g, npix, npoly, rbf, cen, raw,
                       ortho, logdet);
  else
    hipLaunchKernelGGL(basis_build_kernel<64>, dim3(G), dim3(TB_NT), 0,
                       rvs_stream(stream), lam, npix_g, npix, npoly, rbf, cen, raw,
                       ortho, logdet);
  RVS_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
// CCF tables of every grid (ccf_tables.py: rebin_tables, interp_spline_tables
// without the collocation matrix, bin_ranges).  nodes / edges of the continuum
// spline come from the host (make_ccf.py:123-131: a log and an exp per node).
// ---------------------------------------------------------------------------
#define TB_MAXNODE 24
__global__ void __launch_bounds__(TB_NT)
    ccf_tables_kernel(const double *__restrict__ lam,
                      const int32_t *__restrict__ npix_g, int npix_s,
                      const double *__restrict__ ccf_lam, int nfft, int continuum,
                      const double *__restrict__ nodes,
                      const double *__restrict__ edges,
                      const int32_t *__restrict__ nnode_g, int nn_s,
                      int32_t *__restrict__ xind, double *__restrict__ rw,
                      double *__restrict__ Eb, int32_t *__restrict__ El,
                      int32_t *__restrict__ istart,
                      int32_t *__restrict__ bin_start) {
#pragma clang fp contract(off)
  __shared__ double t[TB_MAXNODE + 3];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int n = npix_g ? npix_g[g] : npix_s;
  const double *lg = lam + (int64_t)g * npix_s;
  // rebin: xind = searchsorted(lam, ccf_lam) - 1 where a bracketing pixel pair
  // exists, and the right weight
  for (int j = tid; j < nfft; j += TB_NT) {
    const double c = ccf_lam[j];
    const int xi = tb_search(lg, n, c, false) - 1;
    const bool sub = (xi >= 0) && (xi <= n - 2);
    xind[(int64_t)g * nfft + j] = sub ? xi : -1;
    rw[(int64_t)g * nfft + j] = sub ? (c - lg[xi]) / (lg[xi + 1] - lg[xi]) : 0.0;
  }
  if (!continuum) return;
  const int m = nnode_g[g];
  const double *nd = nodes + (int64_t)g * nn_s;
  const double *ed = edges + (int64_t)g * (nn_s + 1);
  // FITPACK knots for s = 0, k = 2 (fpcurf.f): interior knots at data mid points
  const int nt = m + 3;
  if (tid < nt) {
    double v;
    if (tid < 3)
      v = nd[0];
    else if (tid >= nt - 3)
      v = nd[m - 1];
    else
      v = 0.5 * (nd[tid - 2] + nd[tid - 1]);
    t[tid] = v;
  }
  __syncthreads();
  double *eb = Eb + (int64_t)g * npix_s * 3;
  int32_t *el = El + (int64_t)g * npix_s;
  for (int k = tid; k < npix_s; k += TB_NT) {
    if (k >= n) {
      eb[3 * k] = eb[3 * k + 1] = eb[3 * k + 2] = 0.0;
      el[k] = 0;
      continue;
    }
    const double x = lg[k];
    int l = tb_search(t, nt, x, true) - 1;
    l = min(max(l, 2), nt - 4);   // ext = 0: end pieces extrapolate
    const double tl = t[l], tl1 = t[l + 1], tl2 = t[l + 2], tlm1 = t[l - 1];
    // fpbspl, j = 1 then j = 2 (ccf_tables.interp_spline_tables.basis)
    double f = 1.0 / (tl1 - tl);
    const double a0 = f * (tl1 - x);
    const double a1 = f * (x - tl);
    f = a0 / (tl1 - tlm1);
    const double h0 = f * (tl1 - x);
    double h1 = f * (x - tlm1);
    f = a1 / (tl2 - tl);
    h1 = h1 + f * (tl2 - x);
    const double h2 = f * (x - tl);
    eb[3 * k] = h0;
    eb[3 * k + 1] = h1;
    eb[3 * k + 2] = h2;
    el[k] = l - 2;
  }
  __syncthreads();
  // istart[j]: first pixel of knot interval j (searchsorted(El, j)); m - 2 intervals
  const int nint = m - 2;
  int32_t *is = istart + (int64_t)g * nn_s;
  for (int j = tid; j < nn_s; j += TB_NT) {
    int v = 0;
    if (j < nint) {
      int lo = 0, hi = n;   // first k with El[k] >= j
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (el[mid] < j)
          lo = mid + 1;
        else
          hi = mid;
      }
      v = lo;
    } else if (j == nint) {
      v = n;
    }
    is[j] = v;
  }
  // bins of the binned-median start: [start_j, start_{j+1}), the last closed
  int32_t *bs = bin_start + (int64_t)g * (nn_s + 1);
  for (int j = tid; j <= nn_s; j += TB_NT) {
    int v = 0;
    if (j < m)
      v = tb_search(lg, n, ed[j], false);
    else if (j == m)
      v = tb_search(lg, n, ed[m], true);
    bs[j] = v;
  }
}

extern "C" int rvs_ccf_tables_build(const double *lam, const int32_t *npix_g, int G,
                                    int npix, const double *ccf_lam, int nfft,
                                    int continuum, const double *nodes,
                                    const double *edges, const int32_t *nnode_g,
                                    int nnode, int32_t *xind, double *rw, double *Eb,
                                    int32_t *El, int32_t *istart,
                                    int32_t *bin_start, void *stream) {
  if (G < 1 || npix < 2 || nfft < 2 || !lam || !ccf_lam || !xind || !rw)
    return RVS_E_ARG;
  if (continuum && (nnode < 3 || nnode > TB_MAXNODE || !nodes || !edges ||
                    !nnode_g || !Eb || !El || !istart || !bin_start))
    return RVS_E_ARG;
  hipLaunchKernelGGL(ccf_tables_kernel, dim3(G), dim3(TB_NT), 0, rvs_stream(stream),
                     lam, npix_g, npix, ccf_lam, nfft, continuum, nodes, edges,
                     nnode_g, nnode, xind, rw, Eb, El, istart, bin_start);
  RVS_LAUNCH_CHECK();
  return 0;
}
