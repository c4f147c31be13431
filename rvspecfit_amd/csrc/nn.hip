// nn.hip -- NN template evaluator on the f32-input MFMA of gfx950.
//
// Reference: py/rvspecfit/nn/NNInterpolator.py:14-91 (MLP: Linear+SiLU stack,
// last Linear without activation), :159-171 (Mapper.forward) and
// py/rvspecfit/nn/RVSInterpolator.py:36-42 (float64 exp(clip(.,-300,300))).
//
// This is the one genuinely GEMM-shaped piece of the path: Y[B,N] = act(X[B,K]
// W[N,K]^T + b).  v_mfma_f32_32x32x2_f32 keeps exact f32 products and an f32
// fma chain (no bf16/xf32 rounding), as needed for parity with torch float32.
// Block = 4 waves = a 64x64 output tile (2x2 waves of 32x32), K staged through
// LDS in 32-deep slabs with coalesced global loads; LDS rows padded to 33
// floats so that the fragment reads (32 consecutive rows, same k) are
// conflict-free.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NN_BM 64
#define NN_BN 64
#define NN_BK 32
#define NN_LDK 33

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

// final != 0: write float64 exp(clip(y)) to yout64, else SiLU and f32 to yout32
__global__ void __launch_bounds__(256)
    nn_linear_kernel(const float *__restrict__ X, const float *__restrict__ W,
                     const float *__restrict__ bias, int Bn, int K, int N,
                     int final_layer, float *__restrict__ yout32,
                     double *__restrict__ yout64) {
  __shared__ float Xs[NN_BM * NN_LDK];
  __shared__ float Ws[NN_BN * NN_LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int row0 = blockIdx.y * NN_BM, col0 = blockIdx.x * NN_BN;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  for (int k0 = 0; k0 < K; k0 += NN_BK) {
    __syncthreads();
    // 64 rows x 32 k each for X and W; thread -> (row = e / 32, k = e % 32)
    for (int e = tid; e < NN_BM * NN_BK; e += 256) {
      const int r = e >> 5, kk = e & 31;
      const int gr = row0 + r, gk = k0 + kk;
      Xs[r * NN_LDK + kk] = (gr < Bn && gk < K) ? X[(int64_t)gr * K + gk] : 0.f;
      const int gc = col0 + r;
      Ws[r * NN_LDK + kk] = (gc < N && gk < K) ? W[(int64_t)gc * K + gk] : 0.f;
    }
    __syncthreads();
    const float *xa = Xs + (wr * 32 + (lane & 31)) * NN_LDK + (lane >> 5);
    const float *wb = Ws + (wc * 32 + (lane & 31)) * NN_LDK + (lane >> 5);
#pragma unroll
    for (int kk = 0; kk < NN_BK; kk += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[kk], wb[kk], acc, 0, 0, 0);
  }
  const int col = col0 + wc * 32 + (lane & 31);
  if (col >= N) return;
  const float bv = bias[col];
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int row = row0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row >= Bn) continue;
    const float y = acc[r] + bv;
    if (final_layer) {
      double v = (double)y;
      v = fmin(fmax(v, -300.0), 300.0);
      yout64[(int64_t)row * N + col] = exp(v);
    } else {
      yout32[(int64_t)row * N + col] = y / (1.0f + expf(-y));
    }
  }
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
  hipLaunchKernelGGL(nn_map_kernel, dim3((B * ndim + 255) / 256), dim3(256), 0,
                     st, params, B, ndim, log_mask, M, S, act0);
  RVS_LAUNCH_CHECK();
  float *cur = act0, *nxt = act1;
  for (int l = 0; l < nlayer; l++) {
    const int K = dims[l], N = dims[l + 1];
    const int fin = (l == nlayer - 1);
    dim3 grid((N + NN_BN - 1) / NN_BN, (B + NN_BM - 1) / NN_BM);
    hipLaunchKernelGGL(nn_linear_kernel, grid, dim3(256), 0, st, cur, W[l], b[l],
                       B, K, N, fin, nxt, templ);
    RVS_LAUNCH_CHECK();
    float *t = cur;
    cur = nxt;
    nxt = t;
  }
  return 0;
}
