// Device helpers shared by template.hip and objective.hip (gfx950 only).
#pragma once
#include "common.h"

#define MAXDIM 6

struct GridDesc {
  int ndim;
  int lens[MAXDIM];
  int uoff[MAXDIM];       // offset of dimension d in the concatenated uvecs
  int64_t gstride[MAXDIM];  // C-order strides of idgrid
  double ptp[MAXDIM];
  uint32_t log_mask;
};

// a[d] for a thread-dependent d without putting the array into scratch
template <class T>
__device__ __forceinline__ T sel_dim(const T (&a)[MAXDIM], int d) {
  // (every element through a register first: as selects of loads the compiler turns the
  // chain into ONE load at a selected offset -- which puts the whole descriptor into
  // scratch memory, 168 bytes stored per lane of every block that carries one: 320 MB
  // per launch of the optimiser's cell-search kernel, its whole duration)
  T v = a[0];
  asm volatile("" : "+v"(v));
#pragma unroll
  for (int i = 1; i < MAXDIM; i++) {
    T ai = a[i];
    asm volatile("" : "+v"(ai));
    v = (d == i) ? ai : v;
  }
  return v;
}

// np.searchsorted(u, x, 'right') - 1  == np.digitize(x, u) - 1
__device__ __forceinline__ int cell_index(const double *u, int n, double x) {
  if (!(x == x)) return n - 1;  // NaN sorts to the end
  int lo = 0, hi = n;           // first index with u[idx] > x
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (u[mid] <= x)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo - 1;
}

// LDS-resident result of the polylinear cell search
struct PolyLoc {
  double w[1 << MAXDIM];
  int64_t id[1 << MAXDIM];
  double mp[MAXDIM];
  double x[MAXDIM];
  int pos[MAXDIM];
  int flag[MAXDIM];
  double red_d[16];
  int red_i[16];
  double dist;
  int mode, nearest;
};

// GridInterp.__call__ index work (spec_inter.py:134-194) for one parameter
// vector: mode 0 = polylinear (vertex ids in itertools.product order + weights),
// 1 = nearest neighbour (brute-force first minimum == cKDTree.query on
// ptp-scaled coordinates, :127-132), 2 = non-finite parameters.  Called by ALL
// NT threads of the block (contains barriers).
template <int NT>
__device__ __forceinline__ void poly_locate(PolyLoc &L, const GridDesc &G,
                            const double *__restrict__ prow,
                            const int64_t *__restrict__ idgrid,
                            const double *__restrict__ uvecs,
                            const double *__restrict__ vecs_s, int64_t ngrid) {
  const int tid = threadIdx.x;
  const int nd = G.ndim, nv = 1 << nd;
  // the index work is spread over a few threads so that its dependent loads
  // (binary searches, idgrid look-ups) run side by side instead of in one chain
  if (tid < nd) {
    const int d = tid;
    double v = prow[d];
    if (G.log_mask & (1u << d)) v = log10(v);
    L.mp[d] = v;
    const int len_d = sel_dim(G.lens, d);
    const int p = cell_index(uvecs + sel_dim(G.uoff, d), len_d, v);
    L.pos[d] = p;
    L.flag[d] = ((fabs(v) <= 1.79e308) ? 0 : 1) |
                ((p < 0 || p >= len_d - 1) ? 2 : 0);
  }
  __syncthreads();
  bool finite = true, outsidebox = false;
  for (int d = 0; d < nd; d++) {
    if (L.flag[d] & 1) finite = false;
    if (L.flag[d] & 2) outsidebox = true;
  }
  if (!outsidebox) {
    if (tid < nv) {
      // vertices in itertools.product([0,1]^ndim) order: first dim slowest
      const int v = tid;
      int64_t off = 0;
#pragma unroll
      for (int d = 0; d < MAXDIM; d++) {
        if (d < nd) {
          const int bit = (v >> (nd - 1 - d)) & 1;
          off += (int64_t)(L.pos[d] + bit) * G.gstride[d];
        }
      }
      L.id[v] = idgrid[off];
    }
    // (beside the vertex look-ups when the block has a second wave for it)
    constexpr int XO = (NT > 64) ? 64 : 0;
    if (NT <= 64) __syncthreads();
    if (tid >= XO && tid < XO + nd) {
      const int d = tid - XO;
      const double *u = uvecs + sel_dim(G.uoff, d);
      const int p = L.pos[d];
      L.x[d] = (L.mp[d] - u[p]) / (u[p + 1] - u[p]);
    }
    __syncthreads();
    bool hole = false;
    for (int v = 0; v < nv; v++)
      if (L.id[v] < 0) hole = true;
    if (!hole && tid < nv) {
      const int v = tid;
      double w = 1;
      for (int d = 0; d < nd; d++)
        w *= ((v >> (nd - 1 - d)) & 1) ? L.x[d] : (1 - L.x[d]);
      L.w[v] = w;
    }
    if (tid == 0) L.mode = hole ? 1 : 0;
  } else if (tid == 0) {
    L.mode = finite ? 1 : 2;
  }
  __syncthreads();
  const int mode = L.mode;
  if (mode == 1) {
    double q[MAXDIM];
    // cKDTree.query(p / ptp) (spec_inter.py:130-132): a DIVISION, so that the
    // distances of equidistant nodes tie (or not) exactly as the reference's do
#pragma unroll
    for (int d = 0; d < MAXDIM; d++) q[d] = (d < nd) ? L.mp[d] / G.ptp[d] : 0.0;
    double bd = __builtin_inf();
    int bi = 0x7fffffff;
    for (int64_t g = tid; g < ngrid; g += NT) {
      double d2 = 0;
#pragma unroll
      for (int d = 0; d < MAXDIM; d++) {
        // separately rounded square and sum (no fma): equidistant nodes compare
        // equal and the first one wins
#pragma clang fp contract(off)
        if (d < nd) {
          const double df = vecs_s[g * nd + d] - q[d];
          const double sq = df * df;
          d2 = d2 + sq;
        }
      }
      if (d2 < bd) {
        bd = d2;
        bi = (int)g;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double od = __shfl_xor(bd, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (od < bd || (od == bd && oi < bi)) {
        bd = od;
        bi = oi;
      }
    }
    if ((tid & 63) == 0) {
      L.red_d[tid >> 6] = bd;
      L.red_i[tid >> 6] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < NT / 64; w++)
        if (L.red_d[w] < bd || (L.red_d[w] == bd && L.red_i[w] < bi)) {
          bd = L.red_d[w];
          bi = L.red_i[w];
        }
      L.nearest = bi;
      L.dist = sqrt(bd);
    }
    __syncthreads();
  } else if (tid == 0) {
    L.nearest = (mode == 2) ? 0 : -1;
    L.dist = (mode == 2) ? __builtin_inf() : 0.0;
  }
  __syncthreads();
}

// ---- rotational kernel primitives (spec_fit.py:495-562) --------------------
__device__ __forceinline__ void rot_prim(double x, double eps, double &k0,
                                         double &k1) {
  x = fmin(fmax(x, -1.0), 1.0);
  const double pi = 3.141592653589793;
  const double norm = pi * (1 - eps / 3.0);
  const double c1 = 2 * (1 - eps) / norm;
  const double c2 = (pi / 2.0) * eps / norm;
  const double s = sqrt(1 - x * x);
  k0 = c1 * (0.5 * (x * s + asin(x))) + c2 * (x - x * x * x / 3.0);
  k1 = c1 * (-1.0 / 3.0 * (1 - x * x) * s) +
       c2 * (x * x / 2.0 - x * x * x * x / 4.0);
}

__device__ __forceinline__ double rot_segment(double xa, double xb, double slope,
                                              double icpt, double eps) {
  double a0, a1, b0, b1;
  rot_prim(xb, eps, b0, b1);
  rot_prim(xa, eps, a0, a1);
  return slope * (b1 - a1) + icpt * (b0 - a0);
}

