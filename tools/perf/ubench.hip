// micro-benchmarks: f64 VALU fma rate (with SGPR operand), f64 MFMA 16x16x4 rate, both together
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_valu(double* out, const double* __restrict__ s, int iters) {
  double a[32];
  double x = threadIdx.x * 1e-3;
  for (int i = 0; i < 32; i++) a[i] = i;
  for (int it = 0; it < iters; it++) {
    const double c0 = s[it & 7];   // uniform -> sgpr
#pragma unroll
    for (int i = 0; i < 32; i++) a[i] = fma(c0, x, a[i]);
  }
  double r = 0; for (int i = 0; i < 32; i++) r += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
__global__ void __launch_bounds__(256) k_mfma(double* out, int iters) {
  d4 acc[4]; for (int i = 0; i < 4; i++) acc[i] = (d4){0,0,0,0};
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double r = 0; for (int i = 0; i < 4; i++) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
__global__ void __launch_bounds__(256) k_both(double* out, const double* __restrict__ s, int iters) {
  d4 acc[4]; for (int i = 0; i < 4; i++) acc[i] = (d4){0,0,0,0};
  double v[16]; for (int i = 0; i < 16; i++) v[i] = i;
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
  for (int it = 0; it < iters; it++) {
    const double c0 = s[it & 7];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; q++) v[i * 4 + q] = fma(c0, a, v[i * 4 + q]);
    }
  }
  double r = 0; for (int i = 0; i < 4; i++) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; i++) r += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
int main() {
  double *out, *s; hipMalloc(&out, 8 * 256 * 4096); hipMalloc(&s, 64);
  hipMemset(s, 0, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int nb : {256 * 4, 256 * 8, 256 * 12}) {
    float ms;
    k_valu<<<nb, 256>>>(out, s, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k_valu<<<nb, 256>>>(out, s, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("blocks %d VALU f64 fma: %.2f TFLOP/s\n", nb, (double)nb * 256 * iters * 32 * 2 / (ms * 1e-3) / 1e12);
    k_mfma<<<nb, 256>>>(out, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k_mfma<<<nb, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("blocks %d MFMA f64 16x16x4: %.2f TFLOP/s\n", nb, (double)nb * 4 * iters * 4 * 2048 / (ms * 1e-3) / 1e12);
    k_both<<<nb, 256>>>(out, s, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k_both<<<nb, 256>>>(out, s, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)nb * 4 * iters * 4 * 2048 + (double)nb * 256 * iters * 16 * 2;
    printf("blocks %d both: %.2f TFLOP/s total (mfma part %.2f, valu part %.2f)\n", nb, fl / (ms * 1e-3) / 1e12,
           (double)nb * 4 * iters * 4 * 2048 / (ms * 1e-3) / 1e12, (double)nb * 256 * iters * 16 * 2 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
