// micro-benchmark: is v_fmac_f64 with a DPP row_newbcast operand (gfx90a+ "DPALU DPP")
// issued at the plain v_fmac_f64 rate?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k_plain(double* out, const double* __restrict__ s, int iters) {
  double a[32];
  double x = threadIdx.x * 1e-3;
  for (int i = 0; i < 32; i++) a[i] = i;
  for (int it = 0; it < iters; it++) {
    const double c0 = s[it & 7];
#pragma unroll
    for (int i = 0; i < 32; i++) a[i] = fma(c0, x, a[i]);
  }
  double r = 0; for (int i = 0; i < 32; i++) r += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
#define FD(i, n) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #n " row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(c), "v"(x));
__global__ void __launch_bounds__(256) k_dpp(double* out, const double* __restrict__ s, int iters) {
  double a[32];
  double x = threadIdx.x * 1e-3;
  double c = s[threadIdx.x & 7] + threadIdx.x;
  for (int i = 0; i < 32; i++) a[i] = i;
  for (int it = 0; it < iters; it++) {
    FD(0,0) FD(1,1) FD(2,2) FD(3,3) FD(4,4) FD(5,5) FD(6,6) FD(7,7) FD(8,8) FD(9,9) FD(10,10) FD(11,11) FD(12,12) FD(13,13) FD(14,14) FD(15,15)
    FD(16,0) FD(17,1) FD(18,2) FD(19,3) FD(20,4) FD(21,5) FD(22,6) FD(23,7) FD(24,8) FD(25,9) FD(26,10) FD(27,11) FD(28,12) FD(29,13) FD(30,14) FD(31,15)
  }
  double r = 0; for (int i = 0; i < 32; i++) r += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
int main() {
  double *out, *s; hipMalloc(&out, 8 * 256 * 4096); hipMalloc(&s, 64); hipMemset(s, 0, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000; const int nb = 256 * 8; float ms;
  k_plain<<<nb, 256>>>(out, s, 10); hipDeviceSynchronize();
  hipEventRecord(e0); k_plain<<<nb, 256>>>(out, s, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("plain v_fmac_f64 (sgpr operand): %.2f TFLOP/s\n", (double)nb * 256 * iters * 32 * 2 / (ms * 1e-3) / 1e12);
  k_dpp<<<nb, 256>>>(out, s, 10); hipDeviceSynchronize();
  hipEventRecord(e0); k_dpp<<<nb, 256>>>(out, s, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("v_fmac_f64_dpp row_newbcast: %.2f TFLOP/s\n", (double)nb * 256 * iters * 32 * 2 / (ms * 1e-3) / 1e12);
  double h[4]; hipMemcpy(h, out, 32, hipMemcpyDeviceToHost); printf("%g %g\n", h[0], h[1]);
  return 0;
}
