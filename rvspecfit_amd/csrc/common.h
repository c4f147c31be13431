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
