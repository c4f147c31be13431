// tools/perf/nn_bench.hip -- stand-alone timing of the NN template stage (DESI b arm:
// 4-256-256-256-200-6215, 10 000 parameter vectors); includes the kernel source so
// that variants can be tried with -D flags.
#include "../../rvspecfit_amd/csrc/nn.hip"
#include <cstdio>
// (stand-alone build: the option table of librvsgpu.so is not linked; NN_BENCH_NOPIPE=1
// takes the generic wide-layer kernel)
int rvs_opt(int id) { return id == RVS_OPT_NN_PIPE ? getenv("NN_BENCH_NOPIPE") == nullptr : 1; }
#include <vector>
#include <cstdlib>
int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 10000;
  const int dims[6] = {4, 256, 256, 256, 200, argc > 2 ? atoi(argv[2]) : 6215};
  std::vector<float *> W(5), b(5);
  for (int l = 0; l < 5; l++) {
    std::vector<float> w((size_t)dims[l] * dims[l + 1]), bb(dims[l + 1]);
    for (auto &x : w) x = (rand() / (float)RAND_MAX - 0.5f) * 2.f / sqrtf(dims[l]);
    for (auto &x : bb) x = 0.05f * (rand() / (float)RAND_MAX - 0.5f);
    hipMalloc(&W[l], w.size() * 4); hipMalloc(&b[l], bb.size() * 4);
    hipMemcpy(W[l], w.data(), w.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b[l], bb.data(), bb.size() * 4, hipMemcpyHostToDevice);
  }
  std::vector<double> p((size_t)B * 4);
  for (int i = 0; i < B; i++) { p[4 * i] = 4000 + rand() % 3000; p[4 * i + 1] = 2.5; p[4 * i + 2] = -1; p[4 * i + 3] = 0.2; }
  double *dp, *M, *S, *templ; float *a0, *a1;
  hipMalloc(&dp, p.size() * 8); hipMemcpy(dp, p.data(), p.size() * 8, hipMemcpyHostToDevice);
  double m[4] = {3.7, 2.5, -1, 0.5}, s[4] = {0.15, 1.4, 0.6, 0.3};
  hipMalloc(&M, 32); hipMalloc(&S, 32); hipMemcpy(M, m, 32, hipMemcpyHostToDevice); hipMemcpy(S, s, 32, hipMemcpyHostToDevice);
  hipMalloc(&templ, (size_t)B * dims[5] * 8); hipMalloc(&a0, (size_t)B * 256 * 4); hipMalloc(&a1, (size_t)B * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 6; r++) {
    hipEventRecord(e0);
    int rc = rvs_template_nn(dp, B, 4, 1, M, S, 5, W.data(), b.data(), dims, a0, a1, templ, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rc) { printf("rc %d\n", rc); return 1; }
    if (r) best = fminf(best, ms);
  }
  double fl = 0; for (int l = 0; l < 5; l++) fl += 2.0 * dims[l] * dims[l + 1];
  std::vector<double> h((size_t)B * dims[5]); hipMemcpy(h.data(), templ, h.size() * 8, hipMemcpyDeviceToHost);
  double cs = 0; for (double v : h) cs += v;
  printf("checksum %.10e  ", cs);
  {  // run-to-run determinism and a host f64 evaluation of a few rows
    rvs_template_nn(dp, B, 4, 1, M, S, 5, W.data(), b.data(), dims, a0, a1, templ, nullptr);
    std::vector<double> h2(h.size()); hipMemcpy(h2.data(), templ, h.size() * 8, hipMemcpyDeviceToHost);
    size_t nd = 0; for (size_t i = 0; i < h.size(); i++) nd += h[i] != h2[i];
    std::vector<std::vector<float>> hw(5), hb(5);
    for (int l = 0; l < 5; l++) { hw[l].resize((size_t)dims[l] * dims[l + 1]); hb[l].resize(dims[l + 1]);
      hipMemcpy(hw[l].data(), W[l], hw[l].size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hb[l].data(), b[l], hb[l].size() * 4, hipMemcpyDeviceToHost); }
    double worst = 0;
    const int rows[4] = {0, 1, B / 2 + 7, B - 1};
    for (int ri = 0; ri < 4; ri++) {
      const int r = rows[ri];
      std::vector<double> x(4), y;
      for (int d = 0; d < 4; d++) { double v = p[4 * r + d]; if (d == 0) v = log10(v); x[d] = (v - m[d]) / s[d]; }
      for (int l = 0; l < 5; l++) {
        y.assign(dims[l + 1], 0.0);
        for (int n = 0; n < dims[l + 1]; n++) { double a = hb[l][n]; for (int k = 0; k < dims[l]; k++) a += (double)hw[l][(size_t)n * dims[l] + k] * x[k];
          y[n] = l < 4 ? a / (1 + exp(-a)) : exp(a); }
        x = y;
      }
      for (int n = 0; n < dims[5]; n++) worst = fmax(worst, fabs(h[(size_t)r * dims[5] + n] / x[n] - 1));
    }
    printf("[rerun differs in %zu values; max rel err vs host f64 on 4 rows %.2e] ", nd, worst);
  }
  printf("B %d N %d: %.3f ms  %.1f TFLOP/s f32 MFMA (%.3f of 157.3)  out[0..1] %.6f %.6f\n", B, dims[5], best, B * fl / (best * 1e-3) / 1e12, B * fl / (best * 1e-3) / 1e12 / 157.3, h[0], h[1]);
  return 0;
}
