// template.hip -- template construction kernels for gfx950 (SURVEY rows A3, A5,
// A6, A7): polylinear gather/lerp/exp on a regular n-D grid, rotational
// broadening FIR, natural-cubic-spline construct and the stand-alone evaluator.
//
// Reference: py/rvspecfit/spec_inter.py:62-194, read_grid.py:127-145,
// spec_fit.py:357-407 (getCurTempl), :495-682 (vsini), src/spliner.c:7-108.
#include "common.h"

#include "template_dev.h"
#include "nm_internal.h"

// One 256-thread block per spectrum.  The 2^ndim vertex rows are contiguous
// float32 rows of `dats`: consecutive lanes read consecutive pixels (coalesced
// 4-B loads; rows are re-read by every spectrum sharing a cell and stay in L2 /
// Infinity Cache).  Accumulation is float64 in vertex (itertools.product) order.
__global__ void __launch_bounds__(256)
    polylinear_kernel(const float *__restrict__ dats, int64_t ngrid, int ntp,
                      const int64_t *__restrict__ idgrid,
                      const double *__restrict__ uvecs, GridDesc G,
                      const double *__restrict__ vecs_s, int exp_flag,
                      const double *__restrict__ params, double *__restrict__ templ,
                      double *__restrict__ outside, int32_t *__restrict__ cellinfo,
                      double *__restrict__ weights) {
  __shared__ PolyLoc PL;
  __shared__ double red_m[8];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nd = G.ndim, nv = 1 << nd;
  poly_locate<256>(PL, G, params + (int64_t)b * nd, idgrid, uvecs, vecs_s, ngrid);
  const int mode = PL.mode;
  double *sh_w = PL.w;
  int64_t *sh_id = PL.id;
  const int sh_nearest = PL.nearest;
  const double sh_dist = PL.dist;
  double *out = templ + (int64_t)b * ntp;
  double mx = 0;
  bool anynan = false;
  if (mode == 0) {
    // four consecutive pixels per thread: one 16-byte load per vertex row (rows
    // start on 4-byte boundaries only, hence the dword-aligned vector type)
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const int n4 = ntp & ~3;
    for (int k = 4 * tid; k < n4; k += 4 * 256) {
      double a4[4] = {0, 0, 0, 0};
      for (int v = 0; v < nv; v++) {
        const f4u r = *reinterpret_cast<const f4u *>(dats + sh_id[v] * ntp + k);
        const double wv = sh_w[v];
        a4[0] = fma(wv, (double)r.x, a4[0]);
        a4[1] = fma(wv, (double)r.y, a4[1]);
        a4[2] = fma(wv, (double)r.z, a4[2]);
        a4[3] = fma(wv, (double)r.w, a4[3]);
      }
#pragma unroll
      for (int q = 0; q < 4; q++) out[k + q] = exp_flag ? exp(a4[q]) : a4[q];
    }
    for (int k = n4 + tid; k < ntp; k += 256) {
      double acc = 0;
      for (int v = 0; v < nv; v++)
        acc = fma(sh_w[v], (double)dats[sh_id[v] * ntp + k], acc);
      out[k] = exp_flag ? exp(acc) : acc;
    }
  } else {
    // FF(self.dats[ret]) on a float32 row: numpy evaluates exp in float32
    const float *row = dats + (int64_t)sh_nearest * ntp;
    for (int k = tid; k < ntp; k += 256) {
      // numpy's own float32 exp, operation for operation (common.h)
      const double val = exp_flag ? (double)np_expf(row[k]) : (double)row[k];
      out[k] = val;
      if (!(val == val)) anynan = true;
      mx = fmax(mx, fabs(val));
    }
  }
  if (mode != 0) {
    // MAX_VAL guard (spec_fit.py:392-397): only for outside > 0
    mx = wave_max(mx);
    double nanf = wave_sum(anynan ? 1.0 : 0.0);
    if ((tid & 63) == 0) {
      red_m[tid >> 6] = mx;
      red_m[4 + (tid >> 6)] = nanf;
    }
    __syncthreads();
    if (tid == 0) {
      double m = fmax(fmax(red_m[0], red_m[1]), fmax(red_m[2], red_m[3]));
      const double nn = red_m[4] + red_m[5] + red_m[6] + red_m[7];
      double o = sh_dist;
      if (o > 0 && (m > 1e100 || nn > 0 || isinf(m))) o = __builtin_nan("");
      outside[b] = o;
    }
  } else if (tid == 0) {
    outside[b] = 0.0;
  }
  if (cellinfo && tid == 0) {
    int32_t *ci = cellinfo + (int64_t)b * (2 + nv);
    ci[0] = mode;
    ci[1] = sh_nearest;
    for (int v = 0; v < nv; v++) ci[2 + v] = (mode == 0) ? (int32_t)sh_id[v] : -1;
  }
  if (weights && tid < nv)
    weights[(int64_t)b * nv + tid] = (mode == 0) ? sh_w[tid] : 0.0;
}

extern "C" int rvs_template_polylinear(
    const float *dats, int64_t ngrid, int ntp, const int64_t *idgrid,
    const double *uvecs, const int32_t *lens, int ndim, const double *vecs_s,
    const double *ptp, uint32_t log_mask, int exp_flag,
    const double *params, int B, double *templ, double *outside,
    int32_t *cellinfo, double *weights, void *stream) {
  if (ndim < 1 || ndim > MAXDIM || B < 1 || ntp < 1) return RVS_E_ARG;
  GridDesc G;
  G.ndim = ndim;
  G.log_mask = log_mask;
  int off = 0;
  for (int d = 0; d < ndim; d++) {
    G.lens[d] = lens[d];
    G.uoff[d] = off;
    off += lens[d];
    G.ptp[d] = ptp[d];
  }
  int64_t st = 1;
  for (int d = ndim - 1; d >= 0; d--) {
    G.gstride[d] = st;
    st *= lens[d];
  }
  hipLaunchKernelGGL(polylinear_kernel, dim3(B), dim3(256), 0,
                     rvs_stream(stream), dats, ngrid, ntp, idgrid, uvecs, G,
                     vecs_s, exp_flag, params, templ, outside, cellinfo, weights);
  RVS_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
// A3 on an irregular grid: spec_inter.TriInterp (spec_inter.py:11-59), the
// Delaunay evaluator of `interpolation_type = 'triangulation'` libraries.
// The triangulation itself (simplices + barycentric transforms of the
// reference's scipy.spatial.Delaunay object) is an artefact of template
// preparation; here:
//  tri_locate_kernel: find_simplex by exhaustive search, the way scipy's
//    _find_simplex_bruteforce decides "inside" (all ndim+1 barycentric
//    coordinates within [-eps, 1+eps], eps = 100*DBL_EPSILON).  One LANE per
//    query point, the simplex transform is wave-uniform (scalar cache), the
//    simplices are split over grid.y; the lowest matching simplex id wins
//    (atomicMin), so the result does not depend on the launch shape.  On a
//    shared face several simplices match; the interpolant is continuous there.
//  tri_eval_kernel: b = T (p - r), spec = sum_i b_i dats[simplex_i] in vertex
//    order with separately rounded products (numpy's (dats*b[:,None]).sum(0)),
//    exp; outside = the same blend of `extraflags`; no simplex -> NaN.
// ---------------------------------------------------------------------------
#define TRI_MAXDIM 6

__device__ __forceinline__ double tri_map(double v, int d, uint32_t log_mask) {
  return ((log_mask >> d) & 1u) ? log10(v) : v;
}

__global__ void __launch_bounds__(64)
    tri_locate_kernel(const double *__restrict__ transform, int ns, int nd,
                      uint32_t log_mask, const double *__restrict__ params,
                      int B, int nslice, int32_t *__restrict__ simplex) {
  const int b0 = blockIdx.x * 64 + threadIdx.x;
  const bool active = b0 < B;
  const int b = active ? b0 : B - 1;
  double p[TRI_MAXDIM];
  bool finite = true;
  for (int d = 0; d < nd; d++) {
    p[d] = tri_map(params[(int64_t)b * nd + d], d, log_mask);
    if (!(fabs(p[d]) <= 1.79e308)) finite = false;
  }
  const int s0 = (int)((int64_t)ns * blockIdx.y / nslice);
  const int s1 = (int)((int64_t)ns * (blockIdx.y + 1) / nslice);
  const double eps = 100.0 * 2.220446049250313e-16;
  int found = 0x7fffffff;
  for (int s = s0; s < s1; s++) {
    const double *T = transform + (int64_t)s * (nd + 1) * nd;  // wave-uniform
    const double *r = T + nd * nd;
    bool in = true;
    double sum = 0;
    for (int i = 0; i < nd; i++) {
      double c = 0;
      for (int jj = 0; jj < nd; jj++) c += T[i * nd + jj] * (p[jj] - r[jj]);
      sum += c;
      if (!(c >= -eps && c <= 1 + eps)) in = false;
    }
    const double cl = 1.0 - sum;
    if (!(cl >= -eps && cl <= 1 + eps)) in = false;
    if (in && s < found) found = s;
  }
  if (active && finite && found != 0x7fffffff) atomicMin(&simplex[b], found);
}

__device__ __forceinline__ void
    tri_eval_body(const double *__restrict__ dats, int ntp,
                  const int32_t *__restrict__ simplices,
                  const double *__restrict__ transform,
                  const double *__restrict__ extraflags, int nd,
                  uint32_t log_mask, int exp_flag,
                  const double *__restrict__ params,
                  const int32_t *__restrict__ simplex,
                  double *__restrict__ templ, double *__restrict__ outside,
                  double *__restrict__ weights,
                  const int32_t *__restrict__ live) {
  __shared__ double sh_b[TRI_MAXDIM + 1];
  __shared__ int sh_id[TRI_MAXDIM + 1];
  __shared__ double red_m[8];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (live && b >= live[0]) return;   // (rows nobody waits for: rvs_nm_run)
  const int sx = simplex[b];
  double *out = templ + (int64_t)b * ntp;
  if (sx == 0x7fffffff || sx < 0) {  // TriInterp returns nan (spec_inter.py:45-47)
    for (int k = tid; k < ntp; k += 256) out[k] = __builtin_nan("");
    if (tid == 0) outside[b] = __builtin_nan("");
    if (weights && tid <= nd) weights[(int64_t)b * (nd + 1) + tid] = 0.0;
    return;
  }
  if (tid == 0) {
    const double *T = transform + (int64_t)sx * (nd + 1) * nd;
    const double *r = T + nd * nd;
    double p[TRI_MAXDIM];
    for (int d = 0; d < nd; d++)
      p[d] = tri_map(params[(int64_t)b * nd + d], d, log_mask) - r[d];
    double sum = 0;
    for (int i = 0; i < nd; i++) {
      // numpy dot of a row with (p - r): sequential products and sums
      double c = 0;
      for (int jj = 0; jj < nd; jj++) c = __dadd_rn(c, __dmul_rn(T[i * nd + jj], p[jj]));
      sh_b[i] = c;
      sum = __dadd_rn(sum, c);
    }
    sh_b[nd] = 1.0 - sum;
    for (int i = 0; i <= nd; i++) sh_id[i] = simplices[(int64_t)sx * (nd + 1) + i];
  }
  __syncthreads();
  double mx = 0;
  bool anynan = false;
  for (int k = tid; k < ntp; k += 256) {
    double acc = __dmul_rn(dats[(int64_t)sh_id[0] * ntp + k], sh_b[0]);
    for (int i = 1; i <= nd; i++)
      acc = __dadd_rn(acc, __dmul_rn(dats[(int64_t)sh_id[i] * ntp + k], sh_b[i]));
    const double val = exp_flag ? exp(acc) : acc;
    out[k] = val;
    if (!(val == val)) anynan = true;
    mx = fmax(mx, fabs(val));
  }
  mx = wave_max(mx);
  const double nanf = wave_sum(anynan ? 1.0 : 0.0);
  if ((tid & 63) == 0) {
    red_m[tid >> 6] = mx;
    red_m[4 + (tid >> 6)] = nanf;
  }
  __syncthreads();
  if (tid == 0) {
    double o = __dmul_rn(extraflags[sh_id[0]], sh_b[0]);
    for (int i = 1; i <= nd; i++)
      o = __dadd_rn(o, __dmul_rn(extraflags[sh_id[i]], sh_b[i]));
    // MAX_VAL guard of getCurTempl (spec_fit.py:392-397)
    const double m = fmax(fmax(red_m[0], red_m[1]), fmax(red_m[2], red_m[3]));
    const double nn = red_m[4] + red_m[5] + red_m[6] + red_m[7];
    if (o > 0 && (m > 1e100 || nn > 0 || isinf(m))) o = __builtin_nan("");
    outside[b] = o;
  }
  if (weights && tid <= nd) weights[(int64_t)b * (nd + 1) + tid] = sh_b[tid];
}
__global__ void __launch_bounds__(256)
    tri_eval_kernel(const double *__restrict__ dats, int ntp,
                    const int32_t *__restrict__ simplices,
                    const double *__restrict__ transform,
                    const double *__restrict__ extraflags, int nd,
                    uint32_t log_mask, int exp_flag,
                    const double *__restrict__ params,
                    const int32_t *__restrict__ simplex,
                    double *__restrict__ templ, double *__restrict__ outside,
                    double *__restrict__ weights,
                    const int32_t *__restrict__ live = nullptr) {
  tri_eval_body(dats, ntp, simplices, transform, extraflags, nd, log_mask, exp_flag,
                params, simplex, templ, outside, weights, live);
}

// find_simplex through a uniform bucket grid over the (mapped) parameter space: a
// cell lists, in ascending order, every simplex whose bounding box (grown by 1e-9 of
// the grid's extent: far more than the inside test's 100 eps) overlaps it, so the
// simplices that can contain a point are all in the point's cell and the first one
// that passes scipy's inside test is the LOWEST matching id -- what the exhaustive
// search returns -- after a few dozen tests instead of all 10^4 - 10^6 of a
// PHOENIX-size triangulation.  One lane per query; the lists are built once per
// library on the host (library.tri_buckets).
// (One WAVE per query: the lanes test 64 consecutive entries of the cell's list at a
// time and the lowest lane that passes wins -- the list ascends, so the first chunk
// with a match holds the lowest matching id.  One lane per query walked its list
// entry by entry, a dependent chain of loads per entry: ~1 us each, up to 384 of them
// for a point outside the hull -- 0.5 ms per launch in the optimiser's rounds.)
#define TRI_LOC_WAVES 4
__device__ __forceinline__ void
    tri_locate_bucket_body(const double *__restrict__ transform, int nd,
                           uint32_t log_mask, const double *__restrict__ params,
                           int B, const rvs_tri_buckets &K,
                           int32_t *__restrict__ simplex,
                           const int32_t *__restrict__ live) {
  const int b = blockIdx.x * TRI_LOC_WAVES + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= B || (live && b >= live[0])) return;   // (wave-uniform)
  double p[TRI_MAXDIM];
  bool finite = true;
  int64_t cell = 0;
#pragma unroll
  for (int d = 0; d < TRI_MAXDIM; d++) {   // (static indices: K and p in registers)
    if (d >= nd) break;
    p[d] = tri_map(params[(int64_t)b * nd + d], d, log_mask);
    if (!(fabs(p[d]) <= 1.79e308)) finite = false;
    double c = floor((p[d] - K.lo[d]) * K.inv_w[d]);
    c = fmin(fmax(c, 0.0), (double)(K.n[d] - 1));
    cell = cell * K.n[d] + (finite ? (int64_t)c : 0);
  }
  int found = 0x7fffffff;
  if (finite) {
    const double eps = 100.0 * 2.220446049250313e-16;
    int64_t ncell = 1;
#pragma unroll
    for (int d = 0; d < TRI_MAXDIM; d++)
      if (d < nd) ncell *= K.n[d];
    // the point's own cell, then the list of the simplices too wide for the grid
    // (list number ncell, library.tri_buckets); the lower id of the two finds
    for (int pass = 0; pass < 2; pass++) {
    const int64_t li = pass ? ncell : cell;
    const int e0 = K.cell_start[li], e1 = K.cell_start[li + 1];
    for (int eb = e0; eb < e1; eb += 64) {
      const int e = eb + lane;
      bool in = e < e1;
      int s = 0;
      if (in) {
        s = K.cell_list[e];
        const double *T = transform + (int64_t)s * (nd + 1) * nd;
        const double *r = T + nd * nd;
        double sum = 0;
        for (int i = 0; i < nd; i++) {
          double c = 0;
          for (int jj = 0; jj < nd; jj++) c += T[i * nd + jj] * (p[jj] - r[jj]);
          sum += c;
          if (!(c >= -eps && c <= 1 + eps)) in = false;
        }
        const double cl = 1.0 - sum;
        if (!(cl >= -eps && cl <= 1 + eps)) in = false;
      }
      const unsigned long long m = __ballot(in);
      if (m) {
        found = min(found, __shfl(s, __ffsll((long long)m) - 1, 64));
        break;
      }
    }
    }
  }
  if (lane == 0) simplex[b] = found;
}
__global__ void __launch_bounds__(64 * TRI_LOC_WAVES)
    tri_locate_bucket_kernel(const double *__restrict__ transform, int nd,
                             uint32_t log_mask, const double *__restrict__ params,
                             int B, rvs_tri_buckets K,
                             int32_t *__restrict__ simplex,
                             const int32_t *__restrict__ live) {
  tri_locate_bucket_body(transform, nd, log_mask, params, B, K, simplex, live);
}

extern "C" int rvs_template_tri_buckets(
    const double *dats, int ntp, const int32_t *simplices, const double *transform,
    const double *extraflags, int nsimplex, int ndim, uint32_t log_mask,
    int exp_flag, const rvs_tri_buckets *buckets, const double *params, int B,
    double *templ, double *outside, int32_t *simplex, double *weights,
    void *stream) {
  if (ndim < 1 || ndim > TRI_MAXDIM || B < 1 || ntp < 1 || nsimplex < 1 ||
      !simplex || !buckets || !buckets->cell_start || !buckets->cell_list)
    return RVS_E_ARG;
  for (int d = 0; d < ndim; d++)
    if (buckets->n[d] < 1) return RVS_E_ARG;
  hipStream_t st = rvs_stream(stream);
  hipLaunchKernelGGL(tri_locate_bucket_kernel,
                     dim3((B + TRI_LOC_WAVES - 1) / TRI_LOC_WAVES),
                     dim3(64 * TRI_LOC_WAVES), 0, st, transform, ndim, log_mask, params,
                     B, *buckets, simplex, nullptr);
  hipLaunchKernelGGL(tri_eval_kernel, dim3(B), dim3(256), 0, st, dats, ntp,
                     simplices, transform, extraflags, ndim, log_mask, exp_flag,
                     params, simplex, templ, outside, weights, nullptr);
  RVS_LAUNCH_CHECK();
  return 0;
}

// the template rows of an optimiser round on Delaunay libraries (rvs_nm_run,
// rvs_bfgs_run): the first min(B, live[0]) rows of `params` through every arm's
// triangulation -- the kernels above with the arm as grid.y (two launches per
// evaluation instead of two per arm: in the optimiser's rounds a launch costs as much
// as its work), same values as rvs_template_tri_buckets
struct TriArms {
  rvs_nm_tri_arm a[RVS_MAX_ARMS];
};
__global__ void __launch_bounds__(64 * TRI_LOC_WAVES)
    tri_locate_arms_kernel(TriArms A, int nd, const double *__restrict__ params, int B,
                           const int32_t *__restrict__ live) {
  const rvs_nm_tri_arm &T = A.a[blockIdx.y];
  tri_locate_bucket_body(T.transform, nd, T.log_mask, params, B, T.buckets, T.simplex,
                         live);
}
__global__ void __launch_bounds__(256)
    tri_eval_arms_kernel(TriArms A, int nd, const double *__restrict__ params,
                         const int32_t *__restrict__ live) {
  const rvs_nm_tri_arm &T = A.a[blockIdx.y];
  tri_eval_body(T.dats, T.ntp, T.simplices, T.transform, T.extraflags, nd, T.log_mask,
                T.exp_flag, params, T.simplex, T.templ, T.outside, nullptr, live);
}

int rvs_internal_template_tri_arms_n(const double *params, int B, const int32_t *live,
                                     int ndim, int narm, const rvs_nm_tri_arm *arms,
                                     hipStream_t st) {
  if (ndim < 1 || ndim > TRI_MAXDIM || B < 1 || narm < 1 || narm > RVS_MAX_ARMS || !arms)
    return RVS_E_ARG;
  TriArms A;
  for (int a = 0; a < RVS_MAX_ARMS; a++) {
    A.a[a] = arms[a < narm ? a : 0];
    const rvs_nm_tri_arm &T = A.a[a];
    if (!T.buckets.cell_start || !T.buckets.cell_list || !T.simplex || T.ntp < 1)
      return RVS_E_ARG;
  }
  // find_simplex once per TRIANGULATION: arms whose libraries share one (the arms of a
  // setup are computed on one parameter grid; library.py keeps one device copy per
  // distinct triangulation) also share the simplex ids -- the caller hands such arms the
  // same `simplex` buffer
  TriArms L;
  int nloc = 0;
  for (int a = 0; a < narm; a++) {
    bool seen = false;
    for (int q = 0; q < nloc; q++)
      if (L.a[q].transform == arms[a].transform && L.a[q].simplex == arms[a].simplex &&
          L.a[q].log_mask == arms[a].log_mask)
        seen = true;
    if (!seen) L.a[nloc++] = arms[a];
  }
  for (int a = nloc; a < RVS_MAX_ARMS; a++) L.a[a] = L.a[0];
  hipLaunchKernelGGL(tri_locate_arms_kernel,
                     dim3((B + TRI_LOC_WAVES - 1) / TRI_LOC_WAVES, nloc),
                     dim3(64 * TRI_LOC_WAVES), 0, st, L, ndim, params, B, live);
  hipLaunchKernelGGL(tri_eval_arms_kernel, dim3(B, narm), dim3(256), 0, st, A, ndim,
                     params, live);
  RVS_LAUNCH_CHECK();
  return 0;
}

extern "C" int rvs_template_tri(const double *dats, int ntp,
                                const int32_t *simplices,
                                const double *transform,
                                const double *extraflags, int nsimplex, int ndim,
                                uint32_t log_mask, int exp_flag,
                                const double *params, int B, double *templ,
                                double *outside, int32_t *simplex,
                                double *weights, void *stream) {
  if (ndim < 1 || ndim > TRI_MAXDIM || B < 1 || ntp < 1 || nsimplex < 1 ||
      !simplex)
    return RVS_E_ARG;
  hipStream_t st = rvs_stream(stream);
  if (hipMemsetD32Async((hipDeviceptr_t)simplex, 0x7fffffff, B, st) !=
      hipSuccess)
    return RVS_E_LAUNCH;
  const int groups = (B + 63) / 64;
  int nslice = 2048 / groups;
  nslice = nslice < 1 ? 1 : (nslice > 256 ? 256 : nslice);
  if (nslice > nsimplex) nslice = nsimplex;
  hipLaunchKernelGGL(tri_locate_kernel, dim3(groups, nslice), dim3(64), 0, st,
                     transform, nsimplex, ndim, log_mask, params, B, nslice,
                     simplex);
  hipLaunchKernelGGL(tri_eval_kernel, dim3(B), dim3(256), 0, st, dats, ntp,
                     simplices, transform, extraflags, ndim, log_mask, exp_flag,
                     params, simplex, templ, outside, weights);
  RVS_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
// A6: rotational broadening (spec_fit.py:495-682)
// ---------------------------------------------------------------------------
#define VSINI_MAXTAP 2048  // one-sided taps kept in LDS

__global__ void __launch_bounds__(256)
    vsini_kernel(const double *__restrict__ templ,
                 const double *__restrict__ vsini,
                 const double *__restrict__ outside, double lnstep, double eps,
                 int ntp, double *__restrict__ out,
                 int32_t *__restrict__ status) {
  __shared__ double wpos[VSINI_MAXTAP + 1];
  __shared__ double red[8];
  const int b = blockIdx.x, tid = threadIdx.x;
  const double *in = templ + (int64_t)b * ntp;
  double *o = out + (int64_t)b * ntp;
  const double vs = vsini[b];
  const double R = (vs / RVS_C_KMS) / lnstep;
  bool copy = !(vs > 0) || (R < 1e-9);
  if (outside) {
    const double ov = outside[b];
    if (!(fabs(ov) <= 1.79e308)) copy = true;  // spec_fit.py:398-404
  }
  int kmax = 0;
  if (!copy) {
    kmax = (int)ceil(R + 1);
    if (kmax > VSINI_MAXTAP) {
      if (tid == 0 && status) atomicOr(&status[b], RVS_ST_NONFINITE);
      copy = true;
    }
  }
  if (copy) {
    for (int k = tid; k < ntp; k += 256) o[k] = in[k];
    return;
  }
  // taps k = 0..kmax (compute_vsini_kernel, spec_fit.py:565-625)
  double psum = 0;
  for (int k = tid; k <= kmax; k += 256) {
    double w = 0;
    double lo = fmin(fmax(k / R, -1.0), 1.0), hi = fmin(fmax((k + 1) / R, -1.0), 1.0);
    if (hi > lo) w += rot_segment(lo, hi, -R, 1.0 + k, eps);
    lo = fmin(fmax((k - 1) / R, -1.0), 1.0);
    hi = fmin(fmax(k / R, -1.0), 1.0);
    if (hi > lo) w += rot_segment(lo, hi, R, 1.0 - k, eps);
    wpos[k] = w;
    psum += (k == 0) ? w : 2 * w;
  }
  psum = block_sum<4>(psum, red);
  __syncthreads();
  const double inv = 1.0 / psum;
  // 'same' convolution with zero padding (scipy.signal.convolve mode='same')
  for (int i = tid; i < ntp; i += 256) {
    double acc = 0;
    // same summation order as a direct convolution: ascending input index
    for (int m = -kmax; m <= kmax; m++) {
      const int q = i + m;
      if (q >= 0 && q < ntp) acc = fma(in[q], wpos[m < 0 ? -m : m] * inv, acc);
    }
    o[i] = acc;
  }
}

extern "C" int rvs_vsini_convolve(const double *templ, const double *vsini,
                                  const double *outside, double lnstep,
                                  double eps, int ntp, int B, double *out,
                                  void *stream) {
  if (B < 1 || ntp < 1 || templ == out || !(lnstep > 0)) return RVS_E_ARG;
  hipLaunchKernelGGL(vsini_kernel, dim3(B), dim3(256), 0, rvs_stream(stream),
                     templ, vsini, outside, lnstep, eps, ntp, out,
                     (int32_t *)nullptr);
  RVS_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
// A7: natural cubic spline construct (src/spliner.c:7-60).
//
// The tridiagonal system  h[i-1] z[i-1] + 2(h[i-1]+h[i]) z[i] + h[i] z[i+1] = r[i]
// is solved with the same Thomas elimination, parallelised over 256 threads:
// each thread owns a contiguous chunk of rows, eliminates it assuming a zero
// incoming carry, the 256 chunk carries are propagated serially (the chunk
// transfer factors are products of the Thomas multipliers, |m| ~ 0.27), and
// every thread then corrects its chunk.  The modified diagonal recurrence
// (cc_dash) depends only on the knots and is recomputed per block in LDS.
// Result is the Thomas solution up to floating point re-association.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    spline_construct_kernel(const double *__restrict__ knots,
                            const double *__restrict__ ys, int ntp, int form,
                            double4 *__restrict__ coef) {
  extern __shared__ double sm[];
  const int N = ntp, m = N - 2;  // unknowns z[1..N-2] -> index 0..m-1
  double *cp = sm;               // [m]   modified super-diagonal
  double *dp = sm + m;           // [m]   modified rhs -> then z interior
  double *carry = dp + m;        // [257]
  double *mult = carry + 257;    // [257]
  const int b = blockIdx.x, tid = threadIdx.x;
  const double *y = ys + (int64_t)b * N;
  const int CH = (m + 255) / 256;
  const int i0 = tid * CH, i1 = min(m, i0 + CH);
  // ---- pass 0: cp recurrence (knots only).  cp[i] = h[i+1]/(diag_i - h[i] cp[i-1])
  // is a contraction towards ~0.268, so a chunk can be started 48 rows early
  // from any value and be exact to rounding when it reaches its own rows.
  {
    int s = max(0, i0 - 48);
    double c = 0;
    for (int i = s; i < i1; i++) {
      const double h0 = knots[i + 1] - knots[i], h1 = knots[i + 2] - knots[i + 1];
      const double diag = 2 * (h1 + h0);
      const double den = (i == 0) ? diag : diag - h0 * c;
      c = h1 / den;
      if (i >= i0) cp[i] = c;
    }
  }
  __syncthreads();
  // ---- pass 1: forward elimination of the rhs inside the chunk, zero carry-in
  {
    double d = 0, mu = 1;
    for (int i = i0; i < i1; i++) {
      const double h0 = knots[i + 1] - knots[i], h1 = knots[i + 2] - knots[i + 1];
      const double s0 = (y[i + 1] - y[i]) / h0, s1 = (y[i + 2] - y[i + 1]) / h1;
      const double rhs = 6 * (s1 - s0);
      const double diag = 2 * (h1 + h0);
      const double den = (i == 0) ? diag : diag - h0 * cp[i - 1];
      d = (i == 0) ? rhs / den : (rhs - h0 * d) / den;
      mu = (i == 0) ? 0.0 : mu * (-h0 / den);
      dp[i] = d;
    }
    carry[tid + 1] = d;  // chunk output with zero input
    mult[tid + 1] = mu;
  }
  __syncthreads();
  if (tid == 0) {
    double c = 0;
    carry[0] = 0;
    for (int t = 0; t < 256; t++) {
      if (t * CH < m) c = carry[t + 1] + mult[t + 1] * c;  // non-empty chunk
      carry[t + 1] = c;  // true dp at the end of chunk t
    }
  }
  __syncthreads();
  if (tid > 0) {
    const double cin = carry[tid];
    double mu = 1;
    for (int i = i0; i < i1; i++) {
      const double h0 = knots[i + 1] - knots[i], h1 = knots[i + 2] - knots[i + 1];
      const double diag = 2 * (h1 + h0);
      const double den = diag - h0 * cp[i - 1];
      mu *= (-h0 / den);
      dp[i] += mu * cin;
    }
  }
  __syncthreads();
  // ---- pass 2: back substitution z[i] = dp[i] - cp[i] z[i+1], chunked likewise
  {
    double z = 0, mu = 1;
    for (int i = i1 - 1; i >= i0; i--) {
      z = (i == m - 1) ? dp[i] : dp[i] - cp[i] * z;
      mu = (i == m - 1) ? 0.0 : mu * (-cp[i]);
      dp[i] = z;
    }
    carry[tid] = z;
    mult[tid] = mu;
  }
  __syncthreads();
  if (tid == 0) {
    double c = 0;
    carry[256] = 0;
    for (int t = 255; t >= 0; t--) {
      if (t * CH < m) c = carry[t] + mult[t] * c;  // non-empty chunk
      carry[t] = c;  // true z at the start of chunk t
    }
  }
  __syncthreads();
  if (i1 > i0 && i1 < m) {
    const double cin = carry[tid + 1];  // z at start of the next chunk
    double mu = 1;
    for (int i = i1 - 1; i >= i0; i--) {
      mu *= (-cp[i]);
      dp[i] += mu * cin;
    }
  }
  __syncthreads();
  // ---- coefficients (spliner.c:52-59)
  double4 *cf = coef + (int64_t)b * N;
  for (int i = tid; i < N; i += 256) {
    if (i >= N - 1) {  // padding row; form 1 keeps y there (fast_interp)
      cf[i] = make_double4((form & 1) ? y[i] : 0.0, 0, 0, 0);
      continue;
    }
    const double h = knots[i + 1] - knots[i], hinv = 1.0 / h;
    const double zi = (i == 0) ? 0.0 : dp[i - 1];
    const double zi1 = (i + 1 == N - 1) ? 0.0 : dp[i];
    const double t1 = hinv * (1.0 / 6), t2 = h * (1.0 / 6);
    if (form == 0)  // A, B, C, D of spliner.c:52-59
      cf[i] = make_double4(zi1 * t1, zi * t1, y[i + 1] * hinv - zi1 * t2,
                           y[i] * hinv - zi * t2);
    else  // the same cubic in powers of dl = x - x_i: y + dl (b + dl (c + dl d))
      cf[i] = make_double4(y[i], (y[i + 1] - y[i]) * hinv - t2 * (2 * zi + zi1),
                           0.5 * zi, (zi1 - zi) * t1);
  }
}

// ---------------------------------------------------------------------------
// A7 construct, windowed variant for (log-)uniform knots (form bit 1).
// With neighbouring spacings equal to ~1 % both Thomas recurrences contract by
// ~0.268 per row, so a row's value depends on rows further than 32 away by
// < 1e-18: every thread can start its chunk SW_W rows early from zero and be
// exact to rounding on its own rows -- no carries, no serial section (the
// exact kernel above spends most of its time in two 256-step serial carries).
// The pivots depend on the knots only: rvs_spline_factors computes them once
// per template grid, so the per-template work has no division and each
// recurrence step is one LDS read pair + one fma.
// One block per (segment of <= SW_SEG knots, template); barrier-separated
// passes over LDS (rhs, forward, backward), then coalesced 32-B records.
// factors layout, 5 arrays of ntp doubles:
//   g_u = 1/den_u, e_u = h_u/den_u (e_0 = 0), c_u = h_{u+1}/den_u (c_{m-1} = 0),
//   h_i, 1/h_i            (u < m = ntp-2 unknowns, i < ntp-1 intervals)
// forward: d_u = g_u rhs_u - e_u d_{u-1};  backward: z_u = d_u - c_u z_{u+1}
// ---------------------------------------------------------------------------
#define SW_W 32
#define SW_SEG 3200
#define SW_RMAX (SW_SEG + 2 * SW_W + 8)

__global__ void spline_factors_kernel(const double *__restrict__ knots, int ntp,
                                      double *__restrict__ fac) {
  // one thread: the reference's own elimination order (spliner.c:20-38)
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int m = ntp - 2;
  double *g = fac, *e = fac + ntp, *cc = fac + 2 * ntp, *hh = fac + 3 * ntp,
         *ih = fac + 4 * ntp;
  double c = 0;
  for (int u = 0; u < m; u++) {
    const double h0 = knots[u + 1] - knots[u], h1 = knots[u + 2] - knots[u + 1];
    const double den = 2 * (h1 + h0) - h0 * c;  // c == 0 at u == 0
    const double inv = 1.0 / den;
    c = h1 * inv;
    g[u] = inv;
    e[u] = (u == 0) ? 0.0 : h0 * inv;
    cc[u] = (u == m - 1) ? 0.0 : c;
  }
  for (int u = m; u < ntp; u++) g[u] = e[u] = cc[u] = 0;
  for (int i = 0; i < ntp - 1; i++) {
    const double h = knots[i + 1] - knots[i];
    hh[i] = h;
    ih[i] = 1.0 / h;
  }
  hh[ntp - 1] = ih[ntp - 1] = 0;
}

// the same factors in the objective kernel's chunk order (common.h), behind the 5 ntp:
// records {1/h_u, 1/h_{u+1}, g_u, e_u} of row u = t CH + q at [q][t] (RVS_OBJ_CHMAX x
// RVS_OBJ_NT records of 32 bytes: a thread's row is two 16-byte requests, a wave's
// contiguous), then the backward multipliers in pairs {c_u(2 j), c_u(2 j + 1)} at
// [j][t]; zero where there is no such row
__global__ void spline_factors_chunk_kernel(int ntp, double *__restrict__ fac) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= RVS_OBJ_FT_LEN) return;
  const int m = ntp - 2, CH = rvs_obj_chunk_len(m);
  constexpr int NREC = RVS_OBJ_NT * RVS_OBJ_CHMAX;
  const double *g = fac, *e = fac + ntp, *cc = fac + 2 * ntp, *ih = fac + 4 * ntp;
  int q, t, a;
  if (idx < 4 * NREC) {
    a = idx & 3;
    q = (idx >> 2) / RVS_OBJ_NT;
    t = (idx >> 2) % RVS_OBJ_NT;
  } else {
    const int r = idx - 4 * NREC;
    a = 4;
    t = (r >> 1) % RVS_OBJ_NT;
    q = 2 * ((r >> 1) / RVS_OBJ_NT) + (r & 1);
  }
  const int u = t * CH + q;
  double v = 0.0;
  if (q < CH && u < m)
    v = a == 0 ? ih[u] : a == 1 ? ih[u + 1] : a == 2 ? g[u] : a == 3 ? e[u] : cc[u];
  fac[5 * (int64_t)ntp + idx] = v;
}

extern "C" int64_t rvs_spline_factors_len(int ntp) {
  return ntp < 4 ? 0 : 5 * (int64_t)ntp + RVS_OBJ_FT_LEN;
}

extern "C" int rvs_spline_factors(const double *knots, int ntp, double *factors,
                                  void *stream) {
  if (ntp < 4) return RVS_E_ARG;
  hipLaunchKernelGGL(spline_factors_kernel, dim3(1), dim3(64), 0,
                     rvs_stream(stream), knots, ntp, factors);
  if (ntp <= RVS_OBJ_FT_MAX_NTP)
    hipLaunchKernelGGL(spline_factors_chunk_kernel, dim3((RVS_OBJ_FT_LEN + 255) / 256),
                       dim3(256), 0, rvs_stream(stream), ntp, factors);
  RVS_LAUNCH_CHECK();
  return 0;
}

__global__ void __launch_bounds__(256)
    spline_construct_win_kernel(const double *__restrict__ fac,
                                const double *__restrict__ ys, int ntp, int seg,
                                int form, double4 *__restrict__ coef) {
  __shared__ double dp[SW_RMAX];  // g*rhs -> d -> z
  __shared__ double ec[SW_RMAX];  // e during the forward pass, c afterwards
  const int N = ntp, m = N - 2;
  const double *g = fac, *e = fac + N, *cc = fac + 2 * N, *hh = fac + 3 * N,
               *ih = fac + 4 * N;
  const int b = blockIdx.y, tid = threadIdx.x;
  const int k0 = blockIdx.x * seg, k1 = min(N, k0 + seg);
  // unknown u <-> z at knot u+1; records of knots [k0,k1) need z of knots [k0,k1]
  const int u0 = max(0, k0 - 1), u1 = min(m, k1);
  const int r0 = max(0, u0 - SW_W), r1 = min(m, u1 + SW_W);
  const int nr = r1 - r0;
  const double *y = ys + (int64_t)b * N;
  for (int i = tid; i < nr; i += 256) {
    const int u = r0 + i;
    const double y1 = y[u + 1];
    const double s0 = (y1 - y[u]) * ih[u], s1 = (y[u + 2] - y1) * ih[u + 1];
    dp[i] = 6 * (s1 - s0) * g[u];
    ec[i] = e[u];
  }
  __syncthreads();
  const int CH = (nr + 255) / 256;
  const int a0 = tid * CH, a1 = min(nr, a0 + CH);  // local indices
  double loc[16];
  {
    double d = 0;
    for (int i = max(0, a0 - SW_W); i < min(a0, nr); i++) d = dp[i] - ec[i] * d;
    // own chunk: stored after the barrier so the neighbours' warm-ups read rhs
#pragma unroll
    for (int q = 0; q < 16; q++)
      if (a0 + q < a1) {
        d = dp[a0 + q] - ec[a0 + q] * d;
        loc[q] = d;
      }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; q++)
    if (a0 + q < a1) dp[a0 + q] = loc[q];
  __syncthreads();
  for (int i = tid; i < nr; i += 256) ec[i] = cc[r0 + i];
  __syncthreads();
  {
    double z = 0;
    for (int i = min(nr, a1 + SW_W) - 1; i >= a1; i--) z = dp[i] - ec[i] * z;
#pragma unroll
    for (int q = 15; q >= 0; q--)
      if (a0 + q < a1) {
        z = dp[a0 + q] - ec[a0 + q] * z;
        loc[q] = z;
      }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; q++)
    if (a0 + q < a1) dp[a0 + q] = loc[q];
  __syncthreads();
  // coefficients (spliner.c:52-59)
  double4 *cf = coef + (int64_t)b * N;
  for (int i = k0 + tid; i < k1; i += 256) {
    if (i >= N - 1) {  // padding row; form 1 keeps y there (fast_interp)
      cf[i] = make_double4((form & 1) ? y[i] : 0.0, 0, 0, 0);
      continue;
    }
    const double h = hh[i], hinv = ih[i];
    const double zi = (i == 0) ? 0.0 : dp[i - 1 - r0];
    const double zi1 = (i + 1 == N - 1) ? 0.0 : dp[i - r0];
    const double yi = y[i], yi1 = y[i + 1];
    const double t1 = hinv * (1.0 / 6), t2 = h * (1.0 / 6);
    if ((form & 1) == 0)
      cf[i] = make_double4(zi1 * t1, zi * t1, yi1 * hinv - zi1 * t2,
                           yi * hinv - zi * t2);
    else
      cf[i] = make_double4(yi, (yi1 - yi) * hinv - t2 * (2 * zi + zi1),
                           0.5 * zi, (zi1 - zi) * t1);
  }
}

extern "C" int rvs_spline_construct(const double *knots, const double *ys,
                                    int ntp, int B, int form,
                                    const double *factors, double *coef,
                                    void *stream) {
  if (ntp < 4 || B < 1 || form < 0 || form > 3) return RVS_E_ARG;
  if (form & 2) {
    if (!factors) return RVS_E_ARG;
    const int nseg = (ntp + SW_SEG - 1) / SW_SEG;
    const int seg = (ntp + nseg - 1) / nseg;
    dim3 grid(nseg, B);
    hipLaunchKernelGGL(spline_construct_win_kernel, grid, dim3(256), 0,
                       rvs_stream(stream), factors, ys, ntp, seg, form,
                       reinterpret_cast<double4 *>(coef));
    RVS_LAUNCH_CHECK();
    return 0;
  }
  const size_t shm = sizeof(double) * (2 * (size_t)(ntp - 2) + 2 * 257);
  if (shm > 159 * 1024) return RVS_E_ARG;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)spline_construct_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              159 * 1024);
    (void)hipGetLastError();
    attr_set = true;
  }
  hipLaunchKernelGGL(spline_construct_kernel, dim3(B), dim3(256), shm,
                     rvs_stream(stream), knots, ys, ntp, form,
                     reinterpret_cast<double4 *>(coef));
  RVS_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
// A7: stand-alone evaluator, same arithmetic as src/spliner.c:71-108
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    spline_eval_kernel(const double *__restrict__ knots,
                       const double4 *__restrict__ coef, int ntp, int log_step,
                       const double *__restrict__ evalx, int neval,
                       double *__restrict__ ret, int32_t *__restrict__ pos_out,
                       int32_t *__restrict__ status) {
  const int b = blockIdx.y;
  const double *ex = evalx + (int64_t)b * neval;
  const double x0 = knots[0], xl = knots[ntp - 1];
  int st = 0;
  if (ex[0] < x0 || ex[neval - 1] < x0) st = RVS_ST_SPLINE_RANGE;
  if (ex[0] >= xl || ex[neval - 1] >= xl) st = RVS_ST_SPLINE_RANGE;
  double step, off;
  if (log_step) {
    step = log(knots[1] / x0);
    if (fabs(step - log(knots[2] / knots[1])) > 1e-10) st |= RVS_ST_SPLINE_GRID;
    off = log(x0);
  } else {
    step = knots[1] - x0;
    if (fabs(step - (knots[2] - knots[1])) > 1e-10) st |= RVS_ST_SPLINE_GRID;
    off = x0;
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (st) {
    if (i == 0 && status) atomicOr(&status[b], st);
    if (i < neval) ret[(int64_t)b * neval + i] = __builtin_nan("");
    return;
  }
  if (i >= neval) return;
  const double x = ex[i];
  int p = (int)(((log_step ? log(x) : x) - off) / step);
  if (pos_out) pos_out[(int64_t)b * neval + i] = p;
  p = min(max(p, 0), ntp - 2);
  const double4 c = coef[(int64_t)b * ntp + p];
  const double dl = x - knots[p], dr = knots[p + 1] - x;
  ret[(int64_t)b * neval + i] =
      c.x * dl * dl * dl + c.y * dr * dr * dr + c.z * dl + c.w * dr;
}

extern "C" int rvs_spline_eval(const double *knots, const double *coef, int ntp,
                               int log_step, const double *evalx, int neval,
                               int B, double *ret, int32_t *pos,
                               int32_t *status, void *stream) {
  if (ntp < 3 || neval < 1 || B < 1) return RVS_E_ARG;
  hipLaunchKernelGGL(spline_eval_kernel, dim3((neval + 255) / 256, B), dim3(256),
                     0, rvs_stream(stream), knots,
                     reinterpret_cast<const double4 *>(coef), ntp, log_step,
                     evalx, neval, ret, pos, status);
  RVS_LAUNCH_CHECK();
  return 0;
}
