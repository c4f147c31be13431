// Shared device/host helpers for librvsgpu (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rvsgpu.h"
#include "options.h"

#define RVS_C_KMS 299792.458  // spec_fit.py:23

#define RVS_LAUNCH_CHECK()                         \
  do {                                             \
    if (hipGetLastError() != hipSuccess) return RVS_E_LAUNCH; \
  } while (0)

static inline hipStream_t rvs_stream(void *s) { return (hipStream_t)s; }

// ---- wavefront (64 lanes) reductions -------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Reduction (not all-reduce) on the VALU only: DPP row shifts inside the 16-lane
// rows, row broadcasts across them (the canonical GCN/CDNA sequence); the total
// of the 64 lanes ends in LANE 63.  __shfl_xor compiles to ds_bpermute, which
// goes through the LDS pipe: 12 of them per double -- fine for a few values,
// the bottleneck when a block folds 65 sums in 8 waves.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_get(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 =
      __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, BANK_MASK, false);
  const int hi2 =
      __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, BANK_MASK, false);
  return __hiloint2double(hi2, lo2);
}

__device__ __forceinline__ double wave_sum_to63(double v) {
  double t = v + dpp_get<0x111, 0xf, 0xf>(v);  // row_shr:1
  t += dpp_get<0x112, 0xf, 0xf>(v);            // row_shr:2
  t += dpp_get<0x113, 0xf, 0xf>(v);            // row_shr:3
  t += dpp_get<0x114, 0xf, 0xe>(t);            // row_shr:4, lanes 4..15
  t += dpp_get<0x118, 0xf, 0xc>(t);            // row_shr:8, lanes 8..15
  t += dpp_get<0x142, 0xa, 0xf>(t);            // row_bcast:15 -> rows 1, 3
  t += dpp_get<0x143, 0xc, 0xf>(t);            // row_bcast:31 -> rows 2, 3
  return t;                                    // lane 63: total of the wave
}

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blocks of NW waves; `red` is LDS scratch of >= NW doubles.
// All threads receive the result.  Contains two barriers.
template <int NW>
__device__ __forceinline__ double block_sum(double v, double *red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double t = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) t += red[i];
  return t;
}

// ---- numpy's float32 exp ---------------------------------------------------
// GridInterp.__call__ exponentiates a nearest-neighbour row in FLOAT32
// (spec_inter.py:153-160: np.exp of a float32 array).  numpy's float32 exp is
// not correctly rounded (max 2.52 ulp): on x86 hosts with AVX2+FMA or AVX-512
// (numpy >= 1.17, unchanged through 2.2) it is the algorithm published in
// numpy/core/src/umath/loops_exponent_log.dispatch.c.src, restated here
// operation for operation so that the template -- and with it chi^2 at high
// S/N -- is the reference's bit for bit: k = rint(x log2 e) by the 1.5*2^23
// trick, Cody-Waite reduction r = x - k ln 2 in two fma, exp(r) = P5(r)/Q2(r)
// (Remez coefficients), scaled by 2^k.  tests/test_numpy_expf.py checks this
// sequence against np.exp on the host (bit-identical on 2 000 000 values).
__device__ __forceinline__ float np_expf(float x) {
  if (x != x) return x;
  if (x >= 88.72283935546875f) return __builtin_inff();
  if (x <= -103.97208404541015625f) return 0.0f;
  float q = __fmul_rn(x, 1.44269504088896341f);
  q = __fsub_rn(__fadd_rn(q, 12582912.0f), 12582912.0f);
  float r = __fmaf_rn(q, -6.93145752e-1f, x);
  r = __fmaf_rn(q, -1.42860677e-6f, r);
  float num = __fmaf_rn(5.082762527590693718096e-04f, r, 6.757896990527504603057e-03f);
  num = __fmaf_rn(num, r, 5.114512081637298353406e-02f);
  num = __fmaf_rn(num, r, 2.473615434895520810817e-01f);
  num = __fmaf_rn(num, r, 7.257664613233124478488e-01f);
  num = __fmaf_rn(num, r, 9.999999999980870924916e-01f);
  float den = __fmaf_rn(2.159509375685829852307e-02f, r, -2.742335390411667452936e-01f);
  den = __fmaf_rn(den, r, 1.0f);
  return ldexpf(__fdiv_rn(num, den), (int)q);
}

// Chunk geometry of the objective kernel's spline sweeps (objective.hip): its 512
// threads own CH consecutive rows each.  rvs_spline_factors lays the grid's factors out
// a second time in that order (chunk-transposed: entry [q][t] = row t CH + q), so that
// the kernel's loads of "row a0 + q of thread t" are coalesced and land in the registers
// the sweeps use.  CH is ODD where it can be: lane t's rows start at t CH doubles, and
// 64-bit LDS reads at a lane stride of 2 CH banks reach all 64 banks once per half wave
// only for odd CH.
#define RVS_OBJ_NT 512
#define RVS_OBJ_CHMAX 16
#define RVS_OBJ_FT_MAX_NTP (RVS_OBJ_NT * RVS_OBJ_CHMAX)   // 8192
#define RVS_OBJ_FT_LEN (5 * RVS_OBJ_NT * RVS_OBJ_CHMAX)    // doubles behind the 5 ntp
__host__ __device__ inline int rvs_obj_chunk_len(int m) {
  int ch = (m + RVS_OBJ_NT - 1) / RVS_OBJ_NT;
  if (ch < 12) ch = 12;
  if ((ch | 1) <= RVS_OBJ_CHMAX) ch |= 1;
  return ch;
}
