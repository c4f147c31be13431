// tools/perf/obj_bench.hip -- stand-alone timing of the fused objective kernel
// (rvs_objective_fused) on the three DESI-shaped arms (2751 / 2326 / 2881 px,
// 6215 / 5303 / 6449 knots, a 7^4 library of float32 rows) at J random in-grid
// parameter points.  Includes the kernel source under renamed entry points, so
// variants are -D flags; everything else comes from librvsgpu.so.  Prints ms per
// launch, the time one CU spends per (job, arm) block, and a checksum of the
// outputs (bit identity between variants).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -DOBJ_ONLY_P=10 \
//     -o tools/perf/_bin/obj_bench tools/perf/obj_bench.hip \
//     -Lrvspecfit_amd -l:librvsgpu.so -Wl,-rpath,'$ORIGIN/../../../rvspecfit_amd'
//   tools/perf/_bin/obj_bench [J=3000] [reps=5] [grid=7,7,7,7] [sorted=0] [vsini_max=60]
#define rvs_objective_fused bench_objective_fused
#define rvs_objective_from_template bench_objective_from_template
#define rvs_objective_max_ntp bench_objective_max_ntp
#define rvs_objective_work_size bench_objective_work_size
#define rvs_dbg_read bench_dbg_read
// (the kernels too: librvsgpu.so exports host stubs of the same names, and the
// dynamic linker would bind this binary's launches to ITS kernels)
#define objective_kernel bench_objective_kernel
#define objective_locate_kernel bench_objective_locate_kernel
#define objective_sum_kernel bench_objective_sum_kernel
#define obj_dbg bench_obj_dbg
#include "../../rvspecfit_amd/csrc/objective.hip"
#ifdef OBJ_PIPE_EXPERIMENT
#define objective_pipe_kernel bench_objective_pipe_kernel
#include "experiments/objective_pipe.hip"
#endif
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                        \
  do {                                                               \
    hipError_t e_ = (x);                                             \
    if (e_ != hipSuccess) {                                          \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return 1;                                                      \
    }                                                                \
  } while (0)

template <class T>
static T *to_dev(const std::vector<T> &v) {
  T *d = nullptr;
  if (hipMalloc(&d, v.size() * sizeof(T)) != hipSuccess) return nullptr;
  hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  return d;
}

int main(int argc, char **argv) {
  const int J = argc > 1 ? atoi(argv[1]) : 3000;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  int lens[4] = {7, 7, 7, 7};
  if (argc > 3) sscanf(argv[3], "%d,%d,%d,%d", &lens[0], &lens[1], &lens[2], &lens[3]);
  const int sorted = argc > 4 ? atoi(argv[4]) : 0;
  const double vsmax = argc > 5 ? atof(argv[5]) : 60.0;
  const int P = 10, S = std::min(J, 2000), NARM = 3;
  const int npixs[3] = {2751, 2326, 2881};
  int ntps[3] = {6215, 5303, 6449};
  if (getenv("OBJ_BENCH_NTP"))   // (e.g. 6216,5304,6452: rows that start on 16-byte boundaries)
    sscanf(getenv("OBJ_BENCH_NTP"), "%d,%d,%d", &ntps[0], &ntps[1], &ntps[2]);
  const double lam0[3] = {3600, 5760, 7520}, tl0[3] = {3500, 5660, 7420},
               tl1[3] = {5900, 7720, 9924};
  const int64_t ngrid = (int64_t)lens[0] * lens[1] * lens[2] * lens[3];
  // own generator: rand() is also called from inside the HIP runtime
  uint64_t lcg = 0x9E3779B97F4A7C15ull;
  auto rnd = [&lcg]() {
    lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(lcg >> 11) * (1.0 / 9007199254740992.0);
  };
  // grid axes (teff through log10, as the reference's mapper does)
  const double lo[4] = {3000, 0, -2, 0}, hi[4] = {12000, 5, 0, 1};
  std::vector<double> uvecs;
  double ptp[4];
  for (int d = 0; d < 4; d++) {
    for (int i = 0; i < lens[d]; i++) {
      double v = lo[d] + (hi[d] - lo[d]) * i / (lens[d] - 1);
      uvecs.push_back(d == 0 ? log10(v) : v);
    }
    ptp[d] = (d == 0) ? log10(hi[0]) - log10(lo[0]) : hi[d] - lo[d];
  }
  std::vector<int64_t> idgrid(ngrid);
  std::vector<double> vecs_s(ngrid * 4);
  {
    int off[4] = {0, lens[0], lens[0] + lens[1], lens[0] + lens[1] + lens[2]};
    for (int64_t g = 0; g < ngrid; g++) {
      idgrid[g] = g;
      int64_t r = g;
      int ix[4];
      for (int d = 3; d >= 0; d--) {
        ix[d] = r % lens[d];
        r /= lens[d];
      }
      for (int d = 0; d < 4; d++) vecs_s[g * 4 + d] = uvecs[off[d] + ix[d]] / ptp[d];
    }
  }
  int64_t *d_idgrid = to_dev(idgrid);
  double *d_uvecs = to_dev(uvecs), *d_vecs = to_dev(vecs_s);
  // jobs: random in-grid parameters; `sorted`: in order of their grid cell
  std::vector<double> params((size_t)J * 4), vsini(J), vel(J);
  std::vector<int32_t> jspec(J);
  {
    std::vector<std::pair<int64_t, int>> key(J);
    std::vector<double> p0((size_t)J * 4);
    for (int j = 0; j < J; j++) {
      int64_t cell = 0;
      for (int d = 0; d < 4; d++) {
        const double u = 0.02 + 0.96 * rnd();
        p0[j * 4 + d] = lo[d] + (hi[d] - lo[d]) * u;
        const double m = (d == 0) ? (log10(p0[j * 4]) - log10(lo[0])) / ptp[0] : u;
        cell = cell * lens[d] + std::min(lens[d] - 2, (int)(m * (lens[d] - 1)));
      }
      key[j] = {sorted ? cell : j, j};
    }
    std::sort(key.begin(), key.end());
    for (int j = 0; j < J; j++) {
      const int s = key[j].second;
      for (int d = 0; d < 4; d++) params[j * 4 + d] = p0[s * 4 + d];
      vsini[j] = vsmax * rnd();
      vel[j] = -300 + 600 * rnd();
      jspec[j] = s % S;
    }
  }
  double *d_params = to_dev(params), *d_vsini = to_dev(vsini), *d_vel = to_dev(vel);
  int32_t *d_jspec = to_dev(jspec);
  rvs_objective_arm arms[NARM];
  memset(arms, 0, sizeof(arms));
  for (int a = 0; a < NARM; a++) {
    const int npix = npixs[a], ntp = ntps[a];
    std::vector<double> lam(npix), knots(ntp), polysT((size_t)npix * P);
    for (int k = 0; k < npix; k++) lam[k] = lam0[a] + 0.8 * k;
    const double l0 = log(tl0[a] / (1 + 1000 / RVS_C_KMS));
    const double stp = log(1 + 0.4 / (0.5 * (tl0[a] + tl1[a])));
    for (int i = 0; i < ntp; i++) knots[i] = exp(l0 + stp * i);
    for (int k = 0; k < npix; k++) {
      const double x = -1 + 2.0 * k / (npix - 1);
      for (int i = 0; i < P; i++)
        polysT[(size_t)k * P + i] =
            (i < 3 ? pow(x, i)
                   : exp(-0.5 * pow((x - (-1 + 2.0 * (i - 3) / 6)) * 7, 2)));
    }
    std::vector<float> dats((size_t)ngrid * ntp);
    for (int64_t g = 0; g < ngrid; g++) {
      const double a0 = 0.2 * rnd(), ph = 6.28 * rnd();
      for (int i = 0; i < ntp; i++)
        dats[g * ntp + i] = (float)(a0 + 0.3 * sin(0.01 * i + ph) + 0.02 * (rnd() - 0.5));
    }
    std::vector<double> spec((size_t)S * npix), espec((size_t)S * npix);
    for (size_t i = 0; i < spec.size(); i++) {
      spec[i] = 1.2 + 0.3 * (rnd() - 0.5);
      espec[i] = 0.02 + 0.01 * rnd();
    }
    rvs_objective_arm &A = arms[a];
    A.pt.lam = to_dev(lam);
    A.pt.polysT = to_dev(polysT);
    A.pt.spec = to_dev(spec);
    A.pt.espec = to_dev(espec);
    A.pt.knots = to_dev(knots);
    double *work, *fac;
    CK(hipMalloc(&work, rvs_chisq_work_size(npix, S) * 8));
    CK(hipMalloc(&fac, rvs_spline_factors_len(ntp) * 8));
    int rc = rvs_chisq_prepare(A.pt.lam, A.pt.spec, A.pt.espec, npix, S, knots.data(),
                               1, 0.0, work, nullptr);
    if (rc) return printf("prepare rc %d\n", rc), 1;
    rc = rvs_spline_factors(A.pt.knots, ntp, fac, nullptr);
    if (rc) return printf("factors rc %d\n", rc), 1;
    A.pt.work = work;
    A.pt.npix = npix;
    A.pt.S = S;
    A.pt.ntp = ntp;
    A.pt.log_step = 1;
    A.dats = to_dev(dats);
    A.idgrid = d_idgrid;
    A.uvecs = d_uvecs;
    A.vecs_s = d_vecs;
    A.factors = fac;
    A.ngrid = ngrid;
    A.lnstep = stp;
    for (int d = 0; d < 4; d++) {
      A.ptp[d] = ptp[d];
      A.lens[d] = lens[d];
    }
    A.ntp = ntp;
    A.ndim = 4;
    A.log_mask = 1;
    A.exp_flag = 1;
  }
  void *scratch;
  double *d_out;
  int32_t *d_st;
  CK(hipMalloc(&scratch, rvs_objective_work_size(J, NARM)));
  CK(hipMalloc(&d_out, J * 8));
  CK(hipMalloc(&d_st, J * 4));
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  // the one-block-per-item kernel first (RVS_OBJ_PIPE=0): the values the
  // persistent kernel is compared with
  std::vector<double> out0(J);
  float best0 = 1e30f;
  const bool skip_ref = getenv("OBJ_BENCH_SKIP_REF") != nullptr;   // (counter runs)
  auto env_int = [](const char *n, int d) { return getenv(n) ? atoi(getenv(n)) : d; };
  const int want_sort = env_int("OBJ_BENCH_SORT", 1);
  if (skip_ref) {
    rvs_option_set("obj_sort", want_sort);
  } else {
    setenv("RVS_OBJ_PIPE", "0", 1);
    rvs_option_set("obj_sort", 0);      // first run: the caller's job order
    for (int r = 0; r < reps + 1; r++) {
      hipEventRecord(e0);
      int rc0 = rvs_objective_fused(arms, NARM, P, d_params, vsmax > 0 ? d_vsini : nullptr,
                                    d_jspec, J, d_vel, 1e5, 1 | RVS_OBJ_STATUS_STORE,
                                    scratch, d_out, d_st, nullptr);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (r) best0 = fminf(best0, ms);
      if (rc0) return printf("objective (per-block) rc %d\n", rc0), 1;
    }
    CK(hipDeviceSynchronize());
    hipMemcpy(out0.data(), d_out, J * 8, hipMemcpyDeviceToHost);
    hipMemset(d_out, 0, J * 8);
    setenv("RVS_OBJ_PIPE", getenv("OBJ_BENCH_PIPE") ? getenv("OBJ_BENCH_PIPE") : "1", 1);
    rvs_option_set("obj_sort", want_sort);
  }
  float best = 1e30f, sum = 0;
  int rc = 0;
  for (int r = 0; r < reps + 1; r++) {
    hipEventRecord(e0);
    rc = rvs_objective_fused(arms, NARM, P, d_params, vsmax > 0 ? d_vsini : nullptr,
                             d_jspec, J, d_vel, 1e5, 1 | RVS_OBJ_STATUS_STORE, scratch,
                             d_out, d_st, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r) {
      best = fminf(best, ms);
      sum += ms;
    }
  }
  if (rc) return printf("objective rc %d\n", rc), 1;
  CK(hipDeviceSynchronize());
  std::vector<double> out(J);
  std::vector<int32_t> st(J);
  hipMemcpy(out.data(), d_out, J * 8, hipMemcpyDeviceToHost);
  hipMemcpy(st.data(), d_st, J * 4, hipMemcpyDeviceToHost);
  uint64_t h = 1469598103934665603ull;
  int nbad = 0;
  // checksum in the jobs' ORIGINAL order: a sorted run prints the same value
  {
    std::vector<std::pair<int, double>> byid;  // (not needed unsorted)
    for (int j = 0; j < J; j++) {
      uint64_t b;
      memcpy(&b, &out[j], 8);
      h += b * (uint64_t)(2 * (jspec[j] + 17) + 1) + (uint64_t)st[j];
      if (st[j] || !(out[j] == out[j])) nbad++;
    }
  }
  double maxrel = 0;
  int nnan = 0;
  for (int j = 0; j < J; j++) {
    if (skip_ref) break;
    if (out[j] == out[j] && out0[j] == out0[j])
      maxrel = fmax(maxrel, fabs(out[j] - out0[j]) / fabs(out0[j]));
    else if ((out[j] == out[j]) != (out0[j] == out0[j]))
      nnan++;
  }
  printf("unsorted, per-block kernel: best %.3f ms (%.2f us per block-CU); max rel diff %.3g, NaN mismatches %d\n",
         best0, 1e3 * best0 * 256 / ((double)J * NARM), maxrel, nnan);
  const double nblk = (double)J * NARM;
  printf("J %d grid %d,%d,%d,%d sorted %d sort %d vsini<=%g: best %.3f ms mean %.3f ms  "
         "%.2f us per block-CU  checksum %016llx  flagged %d  out[0..2] %.10g %.10g %.10g\n",
         J, lens[0], lens[1], lens[2], lens[3], sorted, want_sort, vsmax,
         best, sum / reps,
         1e3 * best * 256 / nblk, (unsigned long long)h, nbad, out[0], out[1], out[2]);
#ifdef RVS_OBJ_TIMING
  {
    unsigned long long t[24];
    const int drc = bench_dbg_read(t);
    if (drc) printf("dbg_read rc %d (%s)\n", drc, hipGetErrorString(hipGetLastError()));
    double tot = 0;
    for (int i = 0; i < 18; i++) tot += (double)t[i];
    const char *nm[18] = {"locate", "gather: tail", "vsini", "spline (rest)", "tv+normal",
                          "cholesky", "resid", "model pass", "wave reduce", "fold",
                          "spl: rhs", "spl: forward", "spl: hand-over 1",
                          "spl: backward", "spl: hand-over 2", "gather: row bases",
                          "gather: rot kernel", "gather: loop"};
    // (wall_clock64 ticks of 10 ns, summed by thread 0 of every block of every launch)
    const double nl = (skip_ref ? 1.0 : 2.0) * (reps + 1);
    for (int i = 0; i < 18; i++)
      printf("  %-18s %.3f  %7.2f us per block\n", nm[i], t[i] / tot,
             t[i] * 0.01 / (nl * J * NARM));
    printf("  total %.2f us per block between the first and the last stamp\n",
           tot * 0.01 / (nl * J * NARM));
  }
#endif
  return 0;
}
