// issue cost of single VALU instructions on gfx950 (cycles per wave-instruction at two
// waves per SIMD): a loop of 64 independent copies of one instruction per trip, 8 waves
// per CU on every CU.  hipcc -O3 --offload-arch=gfx950 ubench_ops.hip -o ubench_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int OP>
__global__ void __launch_bounds__(512) k(double *out, int iters) {
  double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4,
         a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float f0 = threadIdx.x * 1e-3f + 1.f, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4,
        f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
  int i0 = threadIdx.x;
  const double c = 1.0000001;
  for (int it = 0; it < iters; it++) {
#define ONE(a, f)                                                                      \
  if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a) : "v"(c));            \
  if (OP == 1) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a) : "v"(f));                \
  if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(c));                \
  if (OP == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(c));                \
  if (OP == 4) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a) : "v"(i0));             \
  if (OP == 5) asm volatile("v_rndne_f64 %0, %1" : "=v"(a) : "v"(a));                  \
  if (OP == 6) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(f) : "v"(a));                \
  if (OP == 7) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f) : "v"(f0));           \
  if (OP == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f) : "v"(f0));      \
  if (OP == 9) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(f) : "v"(i0));       \
  if (OP == 10) asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(i0), "v"(f) : "vcc");  \
  if (OP == 11) asm volatile("v_mov_b32 %0, %1" : "=v"(f) : "v"(i0));                  \
  if (OP == 12) asm volatile("v_exp_f32 %0, %1" : "=v"(f) : "v"(f));                   \
  if (OP == 13) asm volatile("v_rcp_f64 %0, %1" : "=v"(a) : "v"(a));                   \
  if (OP == 14) asm volatile("v_mov_b64 %0, %1" : "=v"(a) : "v"(c));                   \
  if (OP == 15) asm volatile("s_and_b64 s[20:21], s[20:21], exec" : : : "s20", "s21");
    REP8(ONE(a0, f0) ONE(a1, f1) ONE(a2, f2) ONE(a3, f3) ONE(a4, f4) ONE(a5, f5) ONE(a6, f6)
             ONE(a7, f7))
  }
  out[blockIdx.x * 512 + threadIdx.x] =
      a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + i0;
}
template <int OP>
void run(const char *name, double *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4000, nb = 256;
  k<OP><<<nb, 512>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<nb, 512>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 2 waves x iters x 64 instructions
  const double ninst = 2.0 * iters * 64;
  printf("%-16s %.3f ms  %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name,
         ms, ms * 1e6 / ninst, ms * 1e6 / ninst * 2.4);
}
int main() {
  double *out;
  hipMalloc(&out, 8 * 512 * 256);
  run<0>("v_fma_f64", out);
  run<1>("v_cvt_f64_f32", out);
  run<2>("v_mul_f64", out);
  run<3>("v_add_f64", out);
  run<4>("v_ldexp_f64", out);
  run<5>("v_rndne_f64", out);
  run<6>("v_cvt_i32_f64", out);
  run<7>("v_fma_f32", out);
  run<8>("v_cndmask_b32", out);
  run<9>("v_lshl_add_u32", out);
  run<10>("v_cmp_lt_i32", out);
  run<11>("v_mov_b32", out);
  run<12>("v_exp_f32", out);
  run<13>("v_rcp_f64", out);
  run<14>("v_mov_b64", out);
  run<15>("s_and_b64", out);
  return 0;
}
