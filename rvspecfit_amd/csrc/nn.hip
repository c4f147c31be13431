// nn.hip -- NN template evaluator on the f32-input MFMA of gfx950.
//
// Reference: py/rvspecfit/nn/NNInterpolator.py:14-91 (MLP: Linear+SiLU stack,
// last Linear without activation), :159-171 (Mapper.forward) and
// py/rvspecfit/nn/RVSInterpolator.py:36-42 (float64 exp(clip(.,-300,300))).
//
// This is the one genuinely GEMM-shaped piece of the path: Y[B,N] = act(X[B,K]
// W[N,K]^T + b).  v_mfma_f32_32x32x2_f32 keeps exact f32 products and an f32
// fma chain (no bf16/xf32 rounding), as needed for parity with torch float32.
// Tiling and staging: see nn_linear_kernel.
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Block tile 128 x 128 (4 waves, each 64 x 64 = 2 x 2 MFMA tiles of 32 x 32: 64
// accumulator registers), K in slabs of 16 through a double-buffered LDS image.
//  * v_mfma_f32_32x32x2_f32 takes A[m][k] from lane (m, k = lane / 32): the k
//    index is free to be permuted as long as A and B agree, so within a group of
//    8 k's the lower half-wave takes k = g + j and the upper k = g + 4 + j in
//    MFMA j (j < 4): ONE ds_read_b128 per operand tile feeds four MFMAs.
//  * rows are padded to 20 floats: the 16-lane groups of a ds_read_b128 then start
//    on 16 different multiples of four banks (conflict free), stores are 16-B
//    aligned ds_write_b128.
//  * the global loads of slab s + 1 are in flight (registers) while slab s is
//    multiplied; one barrier per slab of 32 MFMAs per wave.
// LDS 40 KB per block: four blocks per CU (the 160 KB of gfx950), 4 waves per
// SIMD, so one block's epilogue (bias, SiLU or the float64 exp) runs on the
// VALU under the other blocks' MFMAs.
#define NN_BM 128
#define NN_BN 128
#ifndef NN_BK
#define NN_BK 16
#endif
#define NN_LDK (NN_BK + 4)

__global__ void __launch_bounds__(256)
    nn_map_kernel(const double *__restrict__ params, int B, int ndim,
                  uint32_t log_mask, const double *__restrict__ M,
                  const double *__restrict__ S, float *__restrict__ x) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * ndim) return;
  const int d = i % ndim;
  // Mapper.forward: float32 input, log10 on log_ids, then (y - M)/S in float64
  float y = (float)params[i];
  if (log_mask & (1u << d)) y = (float)log10((double)y);
  x[i] = (float)(((double)y - M[d]) / S[d]);
}

// OutsideInterpolator.__call__ (nn/RVSInterpolator.py:63-71) at the point
// Mapper.forward (nn/NNInterpolator.py:159-171) makes of the parameters, as
// SpecInterpolator.outsideFlag calls it (spec_inter.py:257-272): the largest
// signed distance to the facets of two convex hulls -- the first two and the
// remaining mapped coordinates -- clamped at zero and squared.  (The reference
// asks two Delaunay triangulations whether the point is inside first; for a
// convex hull that is "no facet distance is positive", which the clamp says.)
// One thread per job; a NaN coordinate gives NaN, like numpy's max.
__device__ __forceinline__ double
    nn_outside_point(const double *__restrict__ params, int j, int ndim,
                     uint32_t log_mask, const double *__restrict__ M,
                     const double *__restrict__ S, int mapped,
                     const double *__restrict__ xeqs, int nfx,
                     const double *__restrict__ yeqs, int nfy) {
  double p[8];
  for (int d = 0; d < ndim; d++) {
    if (mapped) {   // the Mapper's float64 output already
      p[d] = params[(int64_t)j * ndim + d];
    } else {
      float y = (float)params[(int64_t)j * ndim + d];   // as nn_map_kernel
      if (log_mask & (1u << d)) y = (float)log10((double)y);
      p[d] = ((double)y - M[d]) / S[d];
    }
  }
  auto hull = [&](const double *eq, int nf, const double *q, int nq) {
    double best = -__builtin_inf();
    for (int f = 0; f < nf; f++) {
      const double *e = eq + (int64_t)f * (nq + 1);
      double v = e[nq];
      for (int d = 0; d < nq; d++) v = fma(e[d], q[d], v);
      best = (v > best || v != v) ? v : best;   // a NaN sticks
    }
    return best;
  };
  const double dx = hull(xeqs, nfx, p, 2);
  const double dy = hull(yeqs, nfy, p + 2, ndim - 2);
  double m = (dx > dy || dx != dx) ? dx : dy;
  m = (m < 0.0) ? 0.0 : m;
  return m * m;
}
__global__ void __launch_bounds__(256)
    nn_outside_kernel(const double *__restrict__ params, int B, int ndim,
                      uint32_t log_mask, const double *__restrict__ M,
                      const double *__restrict__ S, int mapped,
                      const double *__restrict__ xeqs, int nfx,
                      const double *__restrict__ yeqs, int nfy,
                      double *__restrict__ outside) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= B) return;
  outside[j] = nn_outside_point(params, j, ndim, log_mask, M, S, mapped, xeqs,
                                nfx, yeqs, nfy);
}

// exp(y) for |y| <= 300 in float64: y = n ln 2 + r, |r| <= 0.347, degree-12
// Taylor polynomial of exp(r) (truncation 0.347^13 / 13! = 2e-16), scaled by 2^n
// through the exponent field (n within +-433: never subnormal).  18 fp64
// operations against ~35 of the library exp; within 2 ulp of it.
__device__ __forceinline__ double exp_clip300(double y) {
  const double n = rint(y * 1.4426950408889634074);
  double r = fma(n, -6.93147180369123816490e-01, y);
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = 1.0 / 479001600.0;
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const int hi = __double2hiint(p) + ((int)n << 20);
  return __hiloint2double(hi, __double2loint(p));
}

// The same for NE values side by side: every step of the 22-instruction chain is
// issued for all of them before the next step, so that a dependent fp64 operation
// finds its operand ready (written element by element the compiler reuses one
// register set and the wave waits out the full latency 22 times per value: that,
// not the stores, was two thirds of the wide layer's time, tools/perf/nn_variants.sh
// -DNN_DBG_NOEPI).  Same operations per value: the same bits.
template <int NE>
__device__ __forceinline__ void exp_clip300_n(double (&y)[NE]) {
  double n[NE], r[NE], p[NE];
#pragma unroll
  for (int i = 0; i < NE; i++) n[i] = rint(y[i] * 1.4426950408889634074);
#pragma unroll
  for (int i = 0; i < NE; i++) r[i] = fma(n[i], -6.93147180369123816490e-01, y[i]);
#pragma unroll
  for (int i = 0; i < NE; i++) r[i] = fma(n[i], -1.90821492927058770002e-10, r[i]);
#pragma unroll
  for (int i = 0; i < NE; i++) p[i] = fma(1.0 / 479001600.0, r[i], 1.0 / 39916800.0);
#define NN_EXP_STEP(C)         \
  _Pragma("unroll") for (int i = 0; i < NE; i++) p[i] = fma(p[i], r[i], C);
  NN_EXP_STEP(1.0 / 3628800.0)
  NN_EXP_STEP(1.0 / 362880.0)
  NN_EXP_STEP(1.0 / 40320.0)
  NN_EXP_STEP(1.0 / 5040.0)
  NN_EXP_STEP(1.0 / 720.0)
  NN_EXP_STEP(1.0 / 120.0)
  NN_EXP_STEP(1.0 / 24.0)
  NN_EXP_STEP(1.0 / 6.0)
  NN_EXP_STEP(0.5)
  NN_EXP_STEP(1.0)
  NN_EXP_STEP(1.0)
#undef NN_EXP_STEP
#pragma unroll
  for (int i = 0; i < NE; i++) {
    const int hi = __double2hiint(p[i]) + ((int)n[i] << 20);
    y[i] = __hiloint2double(hi, __double2loint(p[i]));
  }
}

// Y = act(X W^T + b): X [Bn, K], W [N, K] (torch Linear.weight), row-major f32.
// final_layer != 0: float64 exp(clip(y, +-300)) to yout64, else SiLU f32.
// BIG = true:  block tile 128 x 128, waves 2 x 2 of 64 x 64 (the wide last layer);
// BIG = false: block tile  32 x 128, waves 1 x 4 of 32 x 32: the 256-wide hidden
//              layers have 2 column tiles only, and with 128-row tiles 158 blocks
//              would occupy 158 of the 256 CUs with one wave per SIMD.
// (bx, nbx: this block's index among the nbx blocks that walk the tiles of ONE
// matrix product: the whole grid of nn_linear_kernel, one y-slice of the grid of
// nn_linear_group_kernel)
template <bool BIG, bool KVEC, bool FINAL>
__device__ __forceinline__ void
    nn_linear_body(const float *__restrict__ X, const float *__restrict__ W,
                   const float *__restrict__ bias, int Bn, int K, int N,
                   float *__restrict__ yout32, double *__restrict__ yout64,
                   const int bx, const int nbx) {
  constexpr int BM = BIG ? NN_BM : 32;
  constexpr int TI = BIG ? 2 : 1, TJ = BIG ? 2 : 1;   // MFMA tiles per wave
  __shared__ __attribute__((aligned(16))) float lds[2][(BM + NN_BN) * NN_LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = BIG ? (wave >> 1) : 0, wc = BIG ? (wave & 1) : wave;
  // Persistent blocks: a block walks over output tiles (row tiles fastest, so
  // the blocks in flight share a few column tiles of W); the first slab of the
  // next tile is requested before the stores of the current one.  The float64
  // output of the last layer -- 497 MB for a 10 000-spectra DESI arm -- costs
  // 0.12 of that layer's 0.33 ms (measured with the stores compiled out) and did
  // not move under any arrangement tried: one tile per block, persistent blocks,
  // next-tile prefetch ahead of the stores, staggered block starts.
  const int ntr = (Bn + BM - 1) / BM, ntc = (N + NN_BN - 1) / NN_BN;
  int row0 = (bx % ntr) * BM, col0 = (bx / ntr) * NN_BN;
  bool first = true;
  for (int tile = bx; tile < ntr * ntc; tile += nbx) {
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; i++)
#pragma unroll
    for (int j = 0; j < TJ; j++)
#pragma unroll
      for (int q = 0; q < 16; q++) acc[i][j][q] = 0.f;
  // staging: thread -> k quad tid % (BK/4), rows tid / (BK/4) + p * RP
  constexpr int QK = NN_BK / 4, RP = 256 / QK;        // quads per row, rows per pass
  constexpr int PX = (BM + RP - 1) / RP, PW = NN_BN / RP;
  const int sr = tid / QK, sq = (tid % QK) * 4;
  // Rows beyond Bn / N are clamped to the last valid row (their products land in
  // outputs that are never stored); only the K padding of the last slab has to
  // be zero.  With K % 4 == 0 (every layer of the reference's architecture) a
  // fetch is unconditional 16-B loads: no divergent control flow in the loop.
  auto gload = [&](const float *base, int nrows, int r, int k) -> f32x4 {
    const float *p = base + (int64_t)min(r, nrows - 1) * K;
    f32x4 v;
    if (KVEC) {   // K % 4 == 0: 16-B aligned rows (a launch-time property)
      v = *reinterpret_cast<const f32x4 *>(p + ((k + 3 < K) ? k : 0));
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) v[q] = (k + q < K) ? p[k + q] : 0.f;
    }
    return v;
  };
  f32x4 gx[PX], gw[PW];
  bool gin = true;   // the quad fetched last lies inside K
  auto fetch = [&](int k0) {
    gin = !KVEC || (k0 + sq + 3 < K);
#pragma unroll
    for (int p = 0; p < PX; p++)
      if (BM % RP == 0 || p * RP + sr < BM)
        gx[p] = gload(X, Bn, row0 + p * RP + sr, k0 + sq);
#pragma unroll
    for (int p = 0; p < PW; p++) gw[p] = gload(W, N, col0 + p * RP + sr, k0 + sq);
  };
  // (the K padding is zeroed here, AFTER the multiplications of the current
  // slab: a select right behind the loads would make the wave wait for them)
  auto stash = [&](int buf) {
    float *xs = lds[buf], *ws = lds[buf] + BM * NN_LDK;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < PX; p++)
      if (BM % RP == 0 || p * RP + sr < BM)
        *reinterpret_cast<f32x4 *>(xs + (p * RP + sr) * NN_LDK + sq) =
            gin ? gx[p] : z;
#pragma unroll
    for (int p = 0; p < PW; p++)
      *reinterpret_cast<f32x4 *>(ws + (p * RP + sr) * NN_LDK + sq) =
          gin ? gw[p] : z;
  };
  const int nslab = (K + NN_BK - 1) / NN_BK;
  if (first) fetch(0);   // (later tiles: fetched before the previous tile's stores)
  first = false;
  stash(0);
  __syncthreads();
  const int fm = lane & 31, fh = (lane >> 5) * 4;
  constexpr int WROWS = 32 * TI, WCOLS = 32 * TJ;
  for (int sidx = 0; sidx < nslab; sidx++) {
    const int buf = sidx & 1;
    if (sidx + 1 < nslab) fetch((sidx + 1) * NN_BK);
    const float *xs = lds[buf] + (wr * WROWS + fm) * NN_LDK + fh;
    const float *ws = lds[buf] + BM * NN_LDK + (wc * WCOLS + fm) * NN_LDK + fh;
#pragma unroll
    for (int g = 0; g < NN_BK; g += 8) {
      f32x4 a[TI], b[TJ];
#pragma unroll
      for (int i = 0; i < TI; i++)
        a[i] = *reinterpret_cast<const f32x4 *>(xs + i * 32 * NN_LDK + g);
#pragma unroll
      for (int j = 0; j < TJ; j++)
        b[j] = *reinterpret_cast<const f32x4 *>(ws + j * 32 * NN_LDK + g);
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
          for (int j = 0; j < TJ; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q],
                                                             acc[i][j], 0, 0, 0);
    }
    if (sidx + 1 < nslab) stash(buf ^ 1);
    __syncthreads();
  }
  // The first slab of the NEXT tile is requested before this tile's stores:
  // vmcnt counts in order, so loads issued behind the 64 stores would not be
  // usable before every store had been acknowledged.
  const int erow0 = row0, ecol0 = col0;
  if (tile + nbx < ntr * ntc) {
    const int nt = tile + nbx;
    row0 = (nt % ntr) * BM;
    col0 = (nt / ntr) * NN_BN;
    fetch(0);
  }
  // epilogue: bias, then SiLU (float32) or the float64 exp.  Tiles that lie
  // inside the matrix (all but the last row / column of tiles) store without
  // any test: straight-line code, 64 stores in flight per lane.
  auto emit = [&](auto inside_c) {
    constexpr bool INSIDE = decltype(inside_c)::value;
#pragma unroll
    for (int tj = 0; tj < TJ; tj++) {
      const int col = ecol0 + wc * WCOLS + tj * 32 + (lane & 31);
      const bool cok = INSIDE || col < N;
      const float bv = bias[cok ? col : 0];
#pragma unroll
      for (int ti = 0; ti < TI; ti++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = erow0 + wr * WROWS + ti * 32 + (r & 3) + 8 * (r >> 2) +
                          4 * (lane >> 5);
          const float y = acc[ti][tj][r] + bv;
          if (FINAL) {
            double v = (double)y;
            v = fmin(fmax(v, -300.0), 300.0);
            const double e = exp_clip300(v);
            if (INSIDE || (cok && row < Bn)) yout64[(int64_t)row * N + col] = e;
          } else {
            const float a = y / (1.0f + expf(-y));
            if (INSIDE || (cok && row < Bn)) yout32[(int64_t)row * N + col] = a;
          }
        }
      }
    }
  };
  if (erow0 + BM <= Bn && ecol0 + NN_BN <= N)
    emit(std::true_type{});
  else
    emit(std::false_type{});
  }   // tiles
}


// ---------------------------------------------------------------------------
// The wide last layer with its epilogue under the NEXT tile's matrix products
// (round 5).  Same tile (128 x 128, waves 2 x 2 of 64 x 64), same staging and the
// same MFMA order per output as nn_linear_body<true, true, true> -- the values
// are its values bit for bit -- but a finished tile is PARKED in a second
// accumulator set and its 64 float64 exps and 8-byte stores per lane are issued
// four at a time at the top of the K slabs of the block's next tile.
//   Why: gfx950 counts vector loads and stores in one counter (vmcnt) and a mix of
// both pending is unordered for the compiler: a wave that needs the next slab's
// operands waits for EVERY store it has issued.  With the whole epilogue behind a
// tile's last slab that was 64 stores (131 KB per block) acknowledged by HBM
// before the next tile could start: the waves 67 % of their cycles in s_waitcnt,
// the matrix pipe 52 % busy (profiles/r04_d_nn_mfma_counters.json).  Now at most
// eight stores are pending when a wave asks, issued a slab's worth of MFMAs
// (~0.85 us) earlier.
//   The slab loop is unrolled (NSLAB = ceil(K / 16) is a template argument: 13 for
// the reference's 200 principal components), so which accumulator elements a slab
// stores is known at compile time and the code of a tile is straight-line: no
// branch between the exps and the MFMAs they hide under.  Stores are buffer stores
// with the tile's rows as the buffer: the lane's part of the offset is one VGPR
// for the whole kernel, the element's part a scalar, rows behind the matrix fall
// out of the buffer's range, columns behind it get an offset that does (no bounds
// branches, edge tiles take the same code).
// ---------------------------------------------------------------------------
template <int I0, int I1, class F>
__device__ __forceinline__ void nn_static_for(F &&f) {
  if constexpr (I0 < I1) {
    f(std::integral_constant<int, I0>{});
    nn_static_for<I0 + 1, I1>(f);
  }
}
#ifndef NN_PIPE_EPI
#define NN_PIPE_EPI 1   // 0: the wide last layer through nn_linear_kernel<true, ., true>
#endif
#ifndef NN_BIG_MIN
#define NN_BIG_MIN 1024   // 128 x 128 tiles from this many of them up, else 32 x 128
#endif
#ifndef NN_XCD_TILES
#define NN_XCD_TILES 1   // wide last layer: an XCD owns every eighth row tile
#endif
#ifndef NN_ST_AUX
#define NN_ST_AUX 0   // cache policy bits of the epilogue's buffer stores
#endif
template <int NSLAB>
__device__ __forceinline__ void
    nn_final_pipe_body(const float *__restrict__ X, const float *__restrict__ W,
                       const float *__restrict__ bias, int Bn, int K, int N,
                       double *__restrict__ yout64, const int bx, const int nbx) {
  constexpr int BM = NN_BM, TI = 2, TJ = 2;
  __shared__ __attribute__((aligned(16))) float lds[2][(BM + NN_BN) * NN_LDK];
  typedef int v2i_t __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int ntr = (Bn + BM - 1) / BM, ntc = (N + NN_BN - 1) / NN_BN;
  // Which tiles this block walks.  Blocks are dealt to the 8 XCDs round robin, and an
  // XCD's L2 (4 MB) is what serves a block's operand slabs: with the tiles walked in
  // one global order every XCD touches all of X (8 MB at 10 000 rows) between two
  // uses of a row tile, and the 497 MB of output stream through the same L2 -- the
  // slabs then come from beyond it, 2-3 us away, one slab of MFMAs ahead is not
  // enough cover and the waves wait.  XCD-aware (NN_XCD_TILES, from 16 row tiles and
  // a grid that is a multiple of 8): XCD x owns the row tiles x, x + 8, ... (1 MB of
  // X) and its blocks walk (column tile, own row tile) with the rows fastest, so
  // what an XCD reads at any time is its rows of X and a few column tiles of W.
  const bool xcd_tiles = NN_XCD_TILES && ntr >= 16 && (nbx & 7) == 0;
  const int xq = bx & 7;
  const int nrx = xcd_tiles ? (ntr - xq + 7) >> 3 : ntr;       // row tiles walked
  const int tstep = xcd_tiles ? nbx >> 3 : nbx;
  const int tend = nrx * ntc;
  auto tile_rc = [&](int u, int &r0, int &c0) {
    const int c = u / nrx, r = u - c * nrx;
    r0 = (xcd_tiles ? xq + 8 * r : r) * BM;
    c0 = c * NN_BN;
  };
  const int tfirst = xcd_tiles ? bx >> 3 : bx;
  int row0, col0;
  tile_rc(tfirst, row0, col0);
  constexpr int QK = NN_BK / 4, RP = 256 / QK;
  constexpr int PX = BM / RP, PW = NN_BN / RP;
  const int sr = tid / QK, sq = (tid % QK) * 4;
  auto gload = [&](const float *base, int nrows, int r, int k) -> f32x4 {
    const float *p = base + (int64_t)min(r, nrows - 1) * K;
    return *reinterpret_cast<const f32x4 *>(p + ((k + 3 < K) ? k : 0));
  };
  f32x4 gx[PX], gw[PW];
  // Only a tile's LAST slab can reach behind K.  There the lanes whose quad lies
  // behind it keep zeros and do not load (exec mask) -- a select on the loaded
  // values, as nn_linear_body has it behind its slab's products, is hoisted by the
  // scheduler of this straight-line code to right behind the loads, where the wave
  // then waits out their whole latency in every slab.
  auto fetch = [&](int k0, auto last_c) {
    if constexpr (decltype(last_c)::value) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < PX; p++) gx[p] = z;
#pragma unroll
      for (int p = 0; p < PW; p++) gw[p] = z;
      if (k0 + sq + 3 < K) {
#pragma unroll
        for (int p = 0; p < PX; p++) gx[p] = gload(X, Bn, row0 + p * RP + sr, k0 + sq);
#pragma unroll
        for (int p = 0; p < PW; p++) gw[p] = gload(W, N, col0 + p * RP + sr, k0 + sq);
      }
    } else {
#pragma unroll
      for (int p = 0; p < PX; p++) gx[p] = gload(X, Bn, row0 + p * RP + sr, k0 + sq);
#pragma unroll
      for (int p = 0; p < PW; p++) gw[p] = gload(W, N, col0 + p * RP + sr, k0 + sq);
    }
  };
  auto stash = [&](int buf) {
    float *xs = lds[buf], *ws = lds[buf] + BM * NN_LDK;
#pragma unroll
    for (int p = 0; p < PX; p++)
      *reinterpret_cast<f32x4 *>(xs + (p * RP + sr) * NN_LDK + sq) = gx[p];
#pragma unroll
    for (int p = 0; p < PW; p++)
      *reinterpret_cast<f32x4 *>(ws + (p * RP + sr) * NN_LDK + sq) = gw[p];
  };
  const int fm = lane & 31, fh = (lane >> 5) * 4;
  // the parked tile
  f32x16 accP[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; i++)
#pragma unroll
    for (int j = 0; j < TJ; j++)
#pragma unroll
      for (int q = 0; q < 16; q++) accP[i][j][q] = 0.f;
  float pbv[TJ] = {0.f, 0.f};
  int pvo[TJ] = {0, 0};
  // (an empty buffer: the stores of "the tile before the first" go nowhere)
  __amdgpu_buffer_rsrc_t prs =
      __builtin_amdgcn_make_buffer_rsrc((void *)yout64, 0, 0, 0x00020000);
  const int lane_off = (4 * (lane >> 5) * N + (lane & 31)) * 8;
  // elements [4 p, 4 p + 4) of the parked tile, in (tj, ti, r) order
  auto epi_portion = [&](auto p_c) {
    constexpr int PP = decltype(p_c)::value;
#ifdef NN_DBG_NOEPI   // (one store per tile keeps the products alive)
    if (PP != 0) return;
#endif
    double ev[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      constexpr int E0 = PP * 4;
      const int e = E0 + q;
      const int tj = e / 32, ti = (e / 16) % 2, r = e % 16;
      const float y = accP[ti][tj][r] + pbv[tj];
      ev[q] = fmin(fmax((double)y, -300.0), 300.0);
    }
#ifndef NN_DBG_NOEXP   // (tools/perf/nn_variants.sh: what the exps cost)
    exp_clip300_n<4>(ev);
#endif
#pragma unroll
    for (int q = 0; q < 4; q++) {
      constexpr int E0 = PP * 4;
      const int e = E0 + q;
      const int tj = e / 32, ti = (e / 16) % 2, r = e % 16;
      const int soff = ((wr * 64 + ti * 32 + (r & 3) + 8 * (r >> 2)) * N + wc * 64 +
                        tj * 32) * 8;
      v2i_t d;
      d.x = __double2loint(ev[q]);
      d.y = __double2hiint(ev[q]);
#ifndef NN_DBG_NOSTORE   // (tools/perf/nn_variants.sh: what the stores cost)
      __builtin_amdgcn_raw_buffer_store_b64(d, prs, pvo[tj], soff, NN_ST_AUX);
#else
      if (d.x == 0x12345 && d.y == 0x54321)
        __builtin_amdgcn_raw_buffer_store_b64(d, prs, pvo[tj], soff, NN_ST_AUX);
#endif
    }
  };
  bool first = true;
  for (int tile = tfirst; tile < tend; tile += tstep) {
    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int j = 0; j < TJ; j++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[i][j][q] = 0.f;
    // (later tiles: fetched under the previous tile's last slab)
    if (first) fetch(0, std::integral_constant<bool, NSLAB == 1>{});
    first = false;
    stash(0);
    __syncthreads();
    const int erow0 = row0, ecol0 = col0;
    const bool more = tile + tstep < tend;
    nn_static_for<0, NSLAB>([&](auto sidx_c) {
      constexpr int sidx = decltype(sidx_c)::value;
      constexpr int buf = sidx & 1;
      // the next slab's operands first (vmcnt counts loads and stores in order: the
      // wait for them at the slab's end must not have this slab's stores ahead of it)
#ifdef NN_DBG_NOLOAD   // (every slab multiplies the tile's first slab again)
      if (sidx + 1 < NSLAB) {
      } else if (more) {
#else
      if (sidx + 1 < NSLAB) {
        fetch((sidx + 1) * NN_BK, std::integral_constant<bool, sidx + 2 == NSLAB>{});
      } else if (more) {   // the first slab of the block's next tile
#endif
        tile_rc(tile + tstep, row0, col0);
        fetch(0, std::integral_constant<bool, NSLAB == 1>{});
      }
      // this slab's share of the parked tile's epilogue (16 portions over NSLAB
      // slabs: the first 16 - NSLAB slabs take two)
      if constexpr (sidx < 16) epi_portion(std::integral_constant<int, sidx>{});
      if constexpr (sidx + NSLAB < 16)
        epi_portion(std::integral_constant<int, sidx + NSLAB>{});
      const float *xs = lds[buf] + (wr * 64 + fm) * NN_LDK + fh;
      const float *ws = lds[buf] + BM * NN_LDK + (wc * 64 + fm) * NN_LDK + fh;
#pragma unroll
      for (int g = 0; g < NN_BK; g += 8) {
        f32x4 a[TI], b[TJ];
#pragma unroll
        for (int i = 0; i < TI; i++)
          a[i] = *reinterpret_cast<const f32x4 *>(xs + i * 32 * NN_LDK + g);
#pragma unroll
        for (int j = 0; j < TJ; j++)
          b[j] = *reinterpret_cast<const f32x4 *>(ws + j * 32 * NN_LDK + g);
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
          for (int i = 0; i < TI; i++)
#pragma unroll
            for (int j = 0; j < TJ; j++)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q],
                                                               acc[i][j], 0, 0, 0);
      }
      if (sidx + 1 < NSLAB) stash(buf ^ 1);
#ifndef NN_DBG_NOBAR
      __syncthreads();
#endif
    });
    // park the tile
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int j = 0; j < TJ; j++) accP[i][j] = acc[i][j];
    prs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(yout64 + (int64_t)erow0 * N), 0,
        min(Bn - erow0, BM) * N * (int)sizeof(double), 0x00020000);
#pragma unroll
    for (int tj = 0; tj < TJ; tj++) {
      const int col = ecol0 + wc * 64 + tj * 32 + (lane & 31);
      pbv[tj] = bias[min(col, N - 1)];
      // (a column behind the matrix: an offset behind any buffer)
      pvo[tj] = col < N ? lane_off + ecol0 * 8 : (int)0x80000000;
    }
  }
  // the block's last tile
  nn_static_for<0, 16>([&](auto p_c) { epi_portion(p_c); });
}

template <int NSLAB>
__global__ void __launch_bounds__(256, 2)
    nn_final_pipe_kernel(const float *__restrict__ X, const float *__restrict__ W,
                         const float *__restrict__ bias, int Bn, int K, int N,
                         double *__restrict__ yout64) {
  static_assert(NSLAB >= 8, "16 portions, at most two per slab");
  nn_final_pipe_body<NSLAB>(X, W, bias, Bn, K, N, yout64, blockIdx.x, gridDim.x);
}
// the slab counts the pipelined kernel is built for: K = 200 (the reference's
// default number of principal components), 128, 256
#define NN_PIPE_SLABS(F) F(13) F(8) F(16)
static inline bool nn_pipe_has(int K) {
  const int ns = (K + NN_BK - 1) / NN_BK;
  return NN_PIPE_EPI && rvs_opt(RVS_OPT_NN_PIPE) && (K & 3) == 0 &&
         (ns == 13 || ns == 8 || ns == 16);
}

// blocks per CU the register budget is held to: the 128 x 128 tile (BIG: 64
// accumulator registers per lane, two operand fragments, the next slab in flight,
// 64 stores with their addresses) needs ~190 VGPRs; at the 128 of four blocks per CU
// it carried 192-268 B of scratch per lane (round 3)
#ifndef NN_MINB_BIG
#define NN_MINB_BIG 2
#endif
template <bool BIG, bool KVEC, bool FINAL>
__global__ void __launch_bounds__(256, BIG ? NN_MINB_BIG : 4)
    nn_linear_kernel(const float *__restrict__ X, const float *__restrict__ W,
                     const float *__restrict__ bias, int Bn, int K, int N,
                     float *__restrict__ yout32,
                     double *__restrict__ yout64) {
  nn_linear_body<BIG, KVEC, FINAL>(X, W, bias, Bn, K, N, yout32, yout64,
                                   blockIdx.x, gridDim.x);
}

// The last layers of several arms' MLPs (same input rows, same K; own weights
// and widths) in ONE launch: grid.y = arm.  An optimiser round evaluates a few
// hundred rows per arm -- one wave of blocks per launch -- and three dependent
// launches per objective call were as long as the objective kernel itself.
#define NN_MAXARM 4
struct NNLinG {
  const float *X[NN_MAXARM], *W[NN_MAXARM], *bias[NN_MAXARM];
  double *y64[NN_MAXARM];
  int N[NN_MAXARM];
};
template <bool BIG, bool KVEC>
__global__ void __launch_bounds__(256, BIG ? NN_MINB_BIG : 4)
    nn_linear_group_kernel(NNLinG G, int Bn, const int32_t *__restrict__ live,
                           int K) {
  // (rows behind the device count are nobody's: rvs_template_nn_arms_n)
  if (live) Bn = min(Bn, live[0]);
  if (Bn < 1) return;
  const int a = blockIdx.y;
  nn_linear_body<BIG, KVEC, true>(G.X[a], G.W[a], G.bias[a], Bn, K, G.N[a],
                                  nullptr, G.y64[a], blockIdx.x, gridDim.x);
}

template <int NSLAB>
__global__ void __launch_bounds__(256, 2)
    nn_final_pipe_group_kernel(NNLinG G, int Bn, const int32_t *__restrict__ live,
                               int K) {
  if (live) Bn = min(Bn, live[0]);
  if (Bn < 1) return;
  const int a = blockIdx.y;
  nn_final_pipe_body<NSLAB>(G.X[a], G.W[a], G.bias[a], Bn, K, G.N[a], G.y64[a],
                            blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------
// The narrow layers in ONE launch: as separate launches the four hidden layers
// of the reference's architecture (4 -> 256 -> 256 -> 256 -> 200) were latency
// bound -- four dependent 10-40 us launches for 3.7 GFLOP.  Here a block owns 32
// rows (parameter vectors) and walks through the layers with the activations in
// LDS (two [32][260] float images, ping-pong); each of its eight waves computes 32
// output columns (one MFMA tile) and takes its B operand
// -- sixteen consecutive k of one row of W per lane and 32-k chunk -- straight from global memory
// (the weights, 256 KB per layer, live in L2 and are read by every block): no
// staging of W, no barrier inside a layer, the loads of the next chunk in
// flight under the MFMAs of the current one (lower half-wave k = G .. G + 15,
// upper G + 16 .. G + 31: whole 128-byte lines per wave instruction).
// The first layer (K = ndim) is a few FMAs per output and runs on the VALU.
// Requirements (checked by the launcher, else layer by layer): every width of
// the fused layers <= 256 and a multiple of 32 from the second layer on.
// ---------------------------------------------------------------------------
#define NH_LD 260
#define NH_MAXL 6
struct NNHidden {
  const float *W[NH_MAXL];
  const float *b[NH_MAXL];
  int dims[NH_MAXL + 1];
  int nl;
};

__device__ __forceinline__ void
    nn_hidden_body(const double *__restrict__ params, int Bn, int ndim,
                   uint32_t log_mask, const double *__restrict__ M,
                   const double *__restrict__ S, const NNHidden &H,
                   float *__restrict__ yout, float (*act)[32 * NH_LD],
                   float *xin) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * 32;
  // Mapper.forward of the block's 32 rows
  if (tid < 32 * 8) {
    const int r = tid >> 3, d = tid & 7;
    float v = 0.f;
    if (d < ndim) {
      const int gr = min(row0 + r, Bn - 1);
      float y = (float)params[(int64_t)gr * ndim + d];
      if (log_mask & (1u << d)) y = (float)log10((double)y);
      v = (float)(((double)y - M[d]) / S[d]);
    }
    xin[tid] = v;
  }
  __syncthreads();
  // first layer (K = ndim <= 8): a handful of FMAs per output, on the VALU;
  // thread = output column and half of the rows
  {
    const int N = H.dims[1], c = tid & 255, rh = (tid >> 8) * 16;
    float w[8], bv = 0.f;
#pragma unroll
    for (int d = 0; d < 8; d++)
      w[d] = (c < N && d < ndim) ? H.W[0][c * ndim + d] : 0.f;
    if (c < N) bv = H.b[0][c];
    for (int r = rh; r < rh + 16; r++) {
      float y = bv;
#pragma unroll
      for (int d = 0; d < 8; d++) y = fmaf(w[d], xin[r * 8 + d], y);
      act[1][r * NH_LD + c] = c < N ? y / (1.0f + expf(-y)) : 0.f;
    }
  }
  __syncthreads();
  const int fm = lane & 31, fh = (lane >> 5) * 16;
  for (int l = 1; l < H.nl; l++) {
    const int K = H.dims[l], N = H.dims[l + 1];   // K % 32 == 0 (zeros to 256)
    const float *Wl = H.W[l];
    const float *in = act[l & 1];
    float *outp = act[(l + 1) & 1];
    const int col0 = wave * 32;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; q++) acc[q] = 0.f;
    // B fragment: 16 consecutive k of one row of W per lane and 32-k chunk
    // (lower half-wave k = G .. G+15, upper G+16 .. G+31: whole 128-byte lines)
    const int c0 = min(col0 + fm, N - 1);
    const float z0 = col0 + fm < N ? 1.f : 0.f;
    const f32x4 *w0 = reinterpret_cast<const f32x4 *>(Wl + (int64_t)c0 * K + fh);
    f32x4 nb0[4];
#pragma unroll
    for (int q = 0; q < 4; q++) nb0[q] = w0[q];
    const bool wave_live = col0 < N;
    for (int G = 0; G < K && wave_live; G += 32) {
      f32x4 cb0[4], a[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        cb0[q] = nb0[q] * z0;
        a[q] = *reinterpret_cast<const f32x4 *>(in + fm * NH_LD + G + fh + 4 * q);
      }
      if (G + 32 < K) {
#pragma unroll
        for (int q = 0; q < 4; q++) nb0[q] = w0[(G + 32) / 4 + q];
      }
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int e = 0; e < 4; e++)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][e], cb0[q][e], acc, 0, 0, 0);
    }
    const bool last = (l == H.nl - 1);
    {
      const int col = col0 + (lane & 31);
      const bool cok = col < N;
      const float bv = H.b[l][cok ? col : 0];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float y = acc[r] + bv;
        const float a = cok ? y / (1.0f + expf(-y)) : 0.f;
        if (last) {
          if (cok && row0 + row < Bn) yout[(int64_t)(row0 + row) * N + col] = a;
        } else {
          outp[row * NH_LD + col] = a;   // (zeros beyond N: the next K padding)
        }
      }
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(512, 2)
    nn_hidden_kernel(const double *__restrict__ params, int Bn, int ndim,
                     uint32_t log_mask, const double *__restrict__ M,
                     const double *__restrict__ S, NNHidden H,
                     float *__restrict__ yout) {
  __shared__ __attribute__((aligned(16))) float act[2][32 * NH_LD];
  __shared__ float xin[32 * 8];
  nn_hidden_body(params, Bn, ndim, log_mask, M, S, H, yout, act, xin);
}

// the hidden stacks of several arms in one launch (grid.y = arm)
struct NNHiddenG {
  NNHidden H[NN_MAXARM];
  const double *M[NN_MAXARM], *S[NN_MAXARM];
  float *yout[NN_MAXARM];
  uint32_t log_mask[NN_MAXARM];
};
__global__ void __launch_bounds__(512, 2)
    nn_hidden_group_kernel(const double *__restrict__ params, int Bn,
                           const int32_t *__restrict__ live, int ndim,
                           NNHiddenG G) {
  __shared__ __attribute__((aligned(16))) float act[2][32 * NH_LD];
  __shared__ float xin[32 * 8];
  if (live) Bn = min(Bn, live[0]);
  if ((int)blockIdx.x * 32 >= Bn) return;
  const int a = blockIdx.y;
  nn_hidden_body(params, Bn, ndim, G.log_mask[a], G.M[a], G.S[a], G.H[a],
                 G.yout[a], act, xin);
}

extern "C" int rvs_template_nn(const double *params, int B, int ndim,
                               uint32_t log_mask, const double *M,
                               const double *S, int nlayer,
                               const float *const *W, const float *const *b,
                               const int32_t *dims, float *act0, float *act1,
                               double *templ, void *stream) {
  // W, b, dims are HOST arrays of device pointers / sizes; dims has nlayer+1
  // entries (dims[0] == ndim)
  if (B < 1 || nlayer < 1 || ndim < 1 || dims[0] != ndim) return RVS_E_ARG;
  hipStream_t st = rvs_stream(stream);
  // every layer but the last in one launch when the widths allow it
  int lfirst = 0;
  float *cur = act0, *nxt = act1;
  bool fuse = nlayer >= 3 && nlayer - 1 <= NH_MAXL && ndim <= 8;
  for (int l = 0; fuse && l < nlayer - 1; l++)
    if (dims[l + 1] > 256 || (l > 0 && (dims[l] & 31))) fuse = false;
#ifdef NN_NO_FUSE  // (tools/perf/nn_bench.hip: the layer-by-layer path timed beside)
  fuse = false;
#endif
  if (fuse) {
    NNHidden H;
    H.nl = nlayer - 1;
    for (int l = 0; l < H.nl; l++) {
      H.W[l] = W[l];
      H.b[l] = b[l];
    }
    for (int l = 0; l <= H.nl; l++) H.dims[l] = dims[l];
    hipLaunchKernelGGL(nn_hidden_kernel, dim3((B + 31) / 32), dim3(512), 0, st,
                       params, B, ndim, log_mask, M, S, H, act1);
    RVS_LAUNCH_CHECK();
    cur = act1;
    nxt = act0;
    lfirst = nlayer - 1;
  } else {
    hipLaunchKernelGGL(nn_map_kernel, dim3((B * ndim + 255) / 256), dim3(256), 0,
                       st, params, B, ndim, log_mask, M, S, act0);
    RVS_LAUNCH_CHECK();
  }
  for (int l = lfirst; l < nlayer; l++) {
    const int K = dims[l], N = dims[l + 1];
    const int fin = (l == nlayer - 1);
    // the wide layer: 128-row tiles; narrow layers: 32-row tiles so that the
    // launch has several blocks per CU
    const bool big = (int64_t)((N + NN_BN - 1) / NN_BN) * ((B + NN_BM - 1) / NN_BM)
                     >= NN_BIG_MIN;
    const bool kv = (K & 3) == 0;
    const int64_t ntile = (int64_t)((N + NN_BN - 1) / NN_BN) *
                          (big ? (B + NN_BM - 1) / NN_BM : (B + 31) / 32);
    // as many persistent blocks as the CUs hold at once (256 CUs)
    const int64_t nres = 256ll * (big ? NN_MINB_BIG : 4);
    const dim3 grid((unsigned)(ntile < nres ? ntile : nres));
#define NN_LAUNCH(BG, KV, FN)                                                  \
  hipLaunchKernelGGL((nn_linear_kernel<BG, KV, FN>), grid, dim3(256), 0, st,   \
                     cur, W[l], b[l], B, K, N, nxt, templ)
    if (fin && big && nn_pipe_has(K)) {
      switch ((K + NN_BK - 1) / NN_BK) {
#define NN_PIPE_CASE(NS)                                                       \
  case NS:                                                                     \
    hipLaunchKernelGGL((nn_final_pipe_kernel<NS>), grid, dim3(256), 0, st, cur, \
                       W[l], b[l], B, K, N, templ);                            \
    break;
        NN_PIPE_SLABS(NN_PIPE_CASE)
#undef NN_PIPE_CASE
      }
    } else if (fin) {
      if (big && kv) NN_LAUNCH(true, true, true);
      else if (big) NN_LAUNCH(true, false, true);
      else if (kv) NN_LAUNCH(false, true, true);
      else NN_LAUNCH(false, false, true);
    } else {
      if (big && kv) NN_LAUNCH(true, true, false);
      else if (big) NN_LAUNCH(true, false, false);
      else if (kv) NN_LAUNCH(false, true, false);
      else NN_LAUNCH(false, false, false);
    }
#undef NN_LAUNCH
    RVS_LAUNCH_CHECK();
    float *t = cur;
    cur = nxt;
    nxt = t;
  }
  return 0;
}

extern "C" int rvs_nn_outside(const double *params, int B, int ndim,
                              uint32_t log_mask, const double *M,
                              const double *S, int mapped, const double *xeqs,
                              int nfx, const double *yeqs, int nfy,
                              double *outside, void *stream) {
  if (B >= 1 && outside && nfx == 0 && nfy == 0) {
    // a library without hulls has no outside check: zeros, on the caller's stream
    return hipMemsetAsync(outside, 0, sizeof(double) * (size_t)B,
                          rvs_stream(stream)) == hipSuccess
               ? 0
               : RVS_E_LAUNCH;
  }
  if (B < 1 || ndim < 3 || ndim > 8 || nfx < 1 || nfy < 1 || !params ||
      (!mapped && (!M || !S)) || !xeqs || !yeqs || !outside)
    return RVS_E_ARG;
  hipLaunchKernelGGL(nn_outside_kernel, dim3((B + 255) / 256), dim3(256), 0,
                     rvs_stream(stream), params, B, ndim, log_mask, M, S, mapped,
                     xeqs, nfx, yeqs, nfy, outside);
  RVS_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
// rvs_template_nn + rvs_nn_outside for the MLPs of several arms at the same
// rows (an optimiser round): three launches in all when the arms' networks have
// one shape up to the output width (the reference's architecture per arm),
// else arm by arm.
// ---------------------------------------------------------------------------
struct NNOutG {
  const double *M[NN_MAXARM], *S[NN_MAXARM], *xeqs[NN_MAXARM], *yeqs[NN_MAXARM];
  double *out[NN_MAXARM];
  int nfx[NN_MAXARM], nfy[NN_MAXARM];
  uint32_t log_mask[NN_MAXARM];
};
__global__ void __launch_bounds__(256)
    nn_outside_group_kernel(const double *__restrict__ params, int B,
                            const int32_t *__restrict__ live, int ndim,
                            NNOutG G) {
  const int a = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= B || (live && j >= live[0])) return;
  if (!G.xeqs[a]) {
    G.out[a][j] = 0.0;
    return;
  }
  G.out[a][j] = nn_outside_point(params, j, ndim, G.log_mask[a], G.M[a], G.S[a],
                                 0, G.xeqs[a], G.nfx[a], G.yeqs[a], G.nfy[a]);
}

extern "C" int rvs_template_nn_arms(const double *params, int B, int ndim,
                                    int narm, const rvs_nm_nn_arm *arms,
                                    void *stream) {
  return rvs_template_nn_arms_n(params, B, nullptr, ndim, narm, arms, stream);
}

extern "C" int rvs_template_nn_arms_n(const double *params, int B,
                                      const int32_t *live, int ndim, int narm,
                                      const rvs_nm_nn_arm *arms, void *stream) {
  if (B < 1 || narm < 1 || !arms || !params) return RVS_E_ARG;
  hipStream_t st = rvs_stream(stream);
  const rvs_nm_nn_arm &a0 = arms[0];
  const int nl = a0.nlayer;
  bool group = narm > 1 && narm <= NN_MAXARM && nl >= 3 && nl - 1 <= NH_MAXL &&
               ndim <= 8 && a0.dims[0] == ndim;
  for (int a = 0; group && a < narm; a++) {
    if (arms[a].nlayer != nl) group = false;
    for (int l = 0; group && l < nl; l++)   // every width but the output's
      if (arms[a].dims[l] != a0.dims[l]) group = false;
  }
  for (int l = 0; group && l < nl - 1; l++)
    if (a0.dims[l + 1] > 256 || (l > 0 && (a0.dims[l] & 31))) group = false;
  if (group && (a0.dims[nl - 1] & 3)) group = false;
  if (!group) {
    for (int a = 0; a < narm; a++) {
      const rvs_nm_nn_arm &n = arms[a];
      int rc = rvs_template_nn(params, B, ndim, n.log_mask, n.M, n.S, n.nlayer,
                               n.W, n.b, n.dims, n.act0, n.act1, n.templ, stream);
      if (rc) return rc;
      if (n.xeqs) {
        rc = rvs_nn_outside(params, B, ndim, n.log_mask, n.M, n.S, 0, n.xeqs,
                            n.nfx, n.yeqs, n.nfy, n.outside, stream);
        if (rc) return rc;
      } else if (hipMemsetAsync(n.outside, 0, sizeof(double) * (size_t)B, st) !=
                 hipSuccess) {
        return RVS_E_LAUNCH;
      }
    }
    return 0;
  }
  NNHiddenG HG;
  NNLinG LG;
  NNOutG OG;
  int nmax = 0;
  for (int a = 0; a < narm; a++) {
    const rvs_nm_nn_arm &n = arms[a];
    HG.H[a].nl = nl - 1;
    for (int l = 0; l < nl - 1; l++) {
      HG.H[a].W[l] = n.W[l];
      HG.H[a].b[l] = n.b[l];
    }
    for (int l = 0; l <= nl - 1; l++) HG.H[a].dims[l] = n.dims[l];
    HG.M[a] = n.M;
    HG.S[a] = n.S;
    HG.yout[a] = n.act1;
    HG.log_mask[a] = n.log_mask;
    LG.X[a] = n.act1;
    LG.W[a] = n.W[nl - 1];
    LG.bias[a] = n.b[nl - 1];
    LG.y64[a] = n.templ;
    LG.N[a] = n.dims[nl];
    if (n.dims[nl] > nmax) nmax = n.dims[nl];
    OG.M[a] = n.M;
    OG.S[a] = n.S;
    OG.xeqs[a] = n.xeqs;
    OG.yeqs[a] = n.yeqs;
    OG.out[a] = n.outside;
    OG.nfx[a] = n.nfx;
    OG.nfy[a] = n.nfy;
    OG.log_mask[a] = n.log_mask;
  }
  hipLaunchKernelGGL(nn_hidden_group_kernel, dim3((B + 31) / 32, narm), dim3(512),
                     0, st, params, B, live, ndim, HG);
  RVS_LAUNCH_CHECK();
  hipLaunchKernelGGL(nn_outside_group_kernel, dim3((B + 255) / 256, narm),
                     dim3(256), 0, st, params, B, live, ndim, OG);
  RVS_LAUNCH_CHECK();
  {
    const int K = a0.dims[nl - 1];
    const int ntc = (nmax + NN_BN - 1) / NN_BN;
    const bool big = (int64_t)ntc * ((B + NN_BM - 1) / NN_BM) >= NN_BIG_MIN;
    const int64_t ntile = (int64_t)ntc * (big ? (B + NN_BM - 1) / NN_BM
                                              : (B + 31) / 32);
    const int64_t nres = 256ll * (big ? NN_MINB_BIG : 4);
    const dim3 grid((unsigned)(ntile < nres ? ntile : nres), narm);
    if (big && nn_pipe_has(K)) {
      switch ((K + NN_BK - 1) / NN_BK) {
#define NN_PIPE_CASE(NS)                                                        \
  case NS:                                                                      \
    hipLaunchKernelGGL((nn_final_pipe_group_kernel<NS>), grid, dim3(256), 0, st, \
                       LG, B, live, K);                                         \
    break;
        NN_PIPE_SLABS(NN_PIPE_CASE)
#undef NN_PIPE_CASE
      }
    } else if (big)
      hipLaunchKernelGGL((nn_linear_group_kernel<true, true>), grid, dim3(256), 0,
                         st, LG, B, live, K);
    else
      hipLaunchKernelGGL((nn_linear_group_kernel<false, true>), grid, dim3(256),
                         0, st, LG, B, live, K);
    RVS_LAUNCH_CHECK();
  }
  return 0;
}

