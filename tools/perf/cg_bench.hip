// tools/perf/cg_bench.hip -- stand-alone timing of chisq_grid_kernel<10> on one
// DESI-b-shaped arm (2751 px, 6215 knots).  Includes the kernel source so that
// variants can be selected with -D flags; prints ms per launch and a checksum
// of the output (bit-identity between variants).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I include -o /tmp/cg_bench tools/perf/cg_bench.hip
#include "../../rvspecfit_amd/csrc/chisq.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

int main(int argc, char **argv) {
  int S = argc > 1 ? atoi(argv[1]) : 2000;
  int Nv = argc > 2 ? atoi(argv[2]) : 400;
  int reps = argc > 3 ? atoi(argv[3]) : 5;
  int pack = argc > 4 ? atoi(argv[4]) : 0;
  int own = argc > 5 ? atoi(argv[5]) : 1;   // 1: every job has its own template (as in the pipeline)
  const int P = 10, npix = 2751, ntp = 6215;
  std::vector<double> lam(npix), knots(ntp), coef(4ll * ntp), polysT((size_t)npix * P);
  for (int k = 0; k < npix; k++) lam[k] = 3600 + 0.8 * k;
  const double l0 = log(3500.), l1 = log(5900.);
  for (int i = 0; i < ntp; i++) knots[i] = exp(l0 + (l1 - l0) * i / (ntp - 1));
  srand(5);
  auto rnd = []() { return rand() / (double)RAND_MAX; };
  for (int i = 0; i < ntp; i++) {
    coef[4 * i] = 1 + 0.3 * sin(0.01 * i) + 0.05 * rnd();
    coef[4 * i + 1] = 0.02 * (rnd() - 0.5);
    coef[4 * i + 2] = 0.01 * (rnd() - 0.5);
    coef[4 * i + 3] = 0.01 * (rnd() - 0.5);
  }
  for (int k = 0; k < npix; k++) {
    const double x = -1 + 2.0 * k / (npix - 1);
    for (int i = 0; i < P; i++)
      polysT[(size_t)k * P + i] = (i < 3 ? pow(x, i) : exp(-0.5 * pow((x - (-1 + 2.0 * (i - 3) / 6)) * 7, 2))) / sqrt((double)npix);
  }
  std::vector<double> spec((size_t)S * npix), espec((size_t)S * npix), vels((size_t)S * Nv);
  for (size_t i = 0; i < spec.size(); i++) { spec[i] = 1 + 0.1 * (rnd() - 0.5); espec[i] = 0.02 + 0.01 * rnd(); }
  for (int s = 0; s < S; s++) for (int v = 0; v < Nv; v++) vels[(size_t)s * Nv + v] = -1000 + 2000.0 * v / Nv;
  double *d_lam, *d_knots, *d_coef, *d_polys, *d_spec, *d_espec, *d_vels, *d_work, *d_out; int32_t *d_st;
  hipMalloc(&d_lam, npix * 8); hipMalloc(&d_knots, ntp * 8); hipMalloc(&d_coef, 32ll * ntp * (own == 1 ? S : (own == 2 ? 76 : 1)));
  hipMalloc(&d_polys, polysT.size() * 8); hipMalloc(&d_spec, spec.size() * 8); hipMalloc(&d_espec, spec.size() * 8);
  hipMalloc(&d_vels, vels.size() * 8); hipMalloc(&d_out, vels.size() * 8); hipMalloc(&d_st, S * 4);
  const int64_t wsz = rvs_chisq_work_size(npix, S);
  hipMalloc(&d_work, wsz * 8);
  hipMemcpy(d_lam, lam.data(), npix * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_knots, knots.data(), ntp * 8, hipMemcpyHostToDevice);
  for (int t = 0; t < (own == 1 ? S : (own == 2 ? 76 : 1)); t++)
    hipMemcpy((char *)d_coef + 32ll * ntp * t, coef.data(), 32ll * ntp, hipMemcpyHostToDevice);
  hipMemcpy(d_polys, polysT.data(), polysT.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_spec, spec.data(), spec.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_espec, espec.data(), spec.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_vels, vels.data(), vels.size() * 8, hipMemcpyHostToDevice);
  hipMemset(d_st, 0, S * 4);
  int32_t *d_jt; hipMalloc(&d_jt, S * 4); hipMemset(d_jt, 0, S * 4);  // own = 0: every job uses template 0
  const int NT_NODE = 76;   // own = 2: 76 node templates, jobs pick one at random (the pipeline's case)
  if (own == 2) {
    std::vector<int32_t> jt(S);
    for (int s = 0; s < S; s++) jt[s] = rand() % NT_NODE;
    hipMemcpy(d_jt, jt.data(), S * 4, hipMemcpyHostToDevice);
  }
  int rc = rvs_chisq_prepare(d_lam, d_spec, d_espec, npix, S, knots.data(), 1, 0.0, d_work, nullptr);
  if (rc) { printf("prepare rc %d\n", rc); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f, sum = 0;
  for (int r = 0; r < reps + 1; r++) {
    hipEventRecord(e0);
    rc = rvs_chisq_grid(d_lam, d_polys, d_work, npix, P, S, d_knots, d_coef, ntp, own == 1 ? S : (own == 2 ? 76 : 1), 1, nullptr, own == 1 ? nullptr : d_jt, S,
                        d_vels, Nv, Nv, nullptr, 1e5, 0.0, pack, d_out, d_st, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r) { best = fminf(best, ms); sum += ms; }
  }
  if (rc) { printf("grid rc %d\n", rc); return 1; }
  if (argc > 6) {   // three arms: one stream after another vs three streams, separate outputs
    const int ns = atoi(argv[6]);
    hipStream_t st[3]; double *o3[3]; hipEvent_t ev[3];
    for (int a = 0; a < 3; a++) { hipStreamCreateWithFlags(&st[a], hipStreamNonBlocking); hipMalloc(&o3[a], vels.size() * 8); hipEventCreateWithFlags(&ev[a], hipEventDisableTiming); }
    float b3 = 1e30f;
    for (int r = 0; r < reps + 1; r++) {
      hipDeviceSynchronize();
      hipEventRecord(e0, st[0]);
      for (int a = 0; a < 3; a++) {
        hipStream_t s = st[ns == 1 ? 0 : a];
        if (ns > 1 && a) { hipStreamWaitEvent(s, e0, 0); }
        rvs_chisq_grid(d_lam, d_polys, d_work, npix, P, S, d_knots, d_coef, ntp, own ? S : 1, 1, nullptr, own ? nullptr : d_jt, S,
                       d_vels, Nv, Nv, nullptr, 1e5, 0.0, pack, o3[a], d_st, s);
        if (ns > 1 && a) { hipEventRecord(ev[a], s); hipStreamWaitEvent(st[0], ev[a], 0); }
      }
      hipEventRecord(e1, st[0]); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (r) b3 = fminf(b3, ms);
    }
    printf("three arms on %d stream(s): best %.3f ms\n", ns, b3);
  }
#ifdef CG_CLOCK
  {
    unsigned long long c[2];
    hipMemcpyFromSymbol(c, HIP_SYMBOL(cg_clock_dbg), sizeof(c));
    printf("wave lifetime: %llu shader cycles, %llu ref ticks (100 MHz) -> %.1f MHz, %.3f ms; %.1f cycles/pixel\n", c[0], c[1],
           (double)c[0] / c[1] * 100.0, c[1] / 1e5, (double)c[0] / npix);
  }
#endif
  std::vector<double> out(vels.size());
  hipMemcpy(out.data(), d_out, out.size() * 8, hipMemcpyDeviceToHost);
  uint64_t h = 1469598103934665603ull; double tot = 0; int nan = 0;
  for (double v : out) { uint64_t b; memcpy(&b, &v, 8); h = (h ^ b) * 1099511628211ull; if (v == v) tot += v; else nan++; }
  const double fl = (double)S * Nv * npix * 170;
  printf("S %d Nv %d: best %.3f ms avg %.3f ms  %.2f TF(170/px)  checksum %016llx sum %.10e nan %d\n", S, Nv, best, sum / reps,
         fl / (best * 1e-3) / 1e12, (unsigned long long)h, tot, nan);
  return 0;
}
