// Shared device/host helpers for librvsgpu (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rvsgpu.h"

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
