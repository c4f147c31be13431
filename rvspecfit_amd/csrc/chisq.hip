// chisq.hip -- Doppler resample + continuum-marginalised chi^2 (SURVEY rows
// A7-eval, A10, A11, A12, A13) for gfx950.
//
// Reference: py/rvspecfit/spec_fit.py:203-354 (get_chisq0), :707-727 (evalRV),
// :797-989 (get_chisq), :992-1092 (find_best), src/spliner.c:71-108 (evaler).
//
// Layout of the hot kernel (chisq_grid_kernel):
//   * one LANE per velocity, one wave per 64 velocities of one (spectrum,
//     template) job, looping over the observed pixels of the arm;
//   * everything that does not depend on the velocity -- the continuum basis
//     row P_i(k), 1/e_k^2, s_k/e_k^2, the pixel's knot coordinate -- is
//     wave-uniform and is fetched through the scalar cache (s_load) and fed to
//     v_fma_f64 as an SGPR operand: no LDS, no cross-lane traffic;
//   * the spline coefficients of the ~dozen knots a wave touches per pixel
//     come through the vector L1 (one 32-B record per knot);
//   * the (p+1)(p+2)/2 - 1 normal-equation sums live in registers; because a
//     lane owns its velocity there is NO reduction -- the p x p Cholesky is
//     done in-lane, fully unrolled.
// The basis products are exactly the reference's: Minv_ij = sum_k (P_i t/e)(P_j t/e),
// v_i = sum_k (P_i t/e)(s/e); -2logL = logdet + 2 sum log e + (D.D - y.y),
// y = L^-1 v, which equals |D - a^T ST|^2 of spec_fit.py:249/298.
#include "common.h"
#include <type_traits>

// ---------------------------------------------------------------------------
// work buffer layout (doubles): [0, G*npix) pixel knot coordinate
//   (log lam_k - log x0)/logstep   (or (lam_k - x0)/step for linear knots)
//   of each of the G wavelength grids of the arm (G = 1: one grid shared by all
//   spectra; else spectrum s is observed on grid grid_id[s], spec_fit.py:70-145
//   takes any `lam` per object)
// [G*npix, G*npix + 2*S*npix) {1/e^2, s/e^2} per spectrum pixel
// then [2*S] {sum log e, sum s^2/e^2}
// then [2*S*npix] {1/e, s/e} per spectrum pixel (ABI 8): the optimiser's objective
//   works in units of sigma and used to form both -- a square root, a division and
//   a product per pixel -- in every one of its ~1100 evaluations per spectrum
// A grid shorter than npix is padded: lam repeats its last value, the spectra on
// it carry espec = +inf there -- such a pixel has weight 0 and is left out of
// sum log e.
// ---------------------------------------------------------------------------
extern "C" int64_t rvs_chisq_work_size_g(int npix, int S, int G) {
  // pix [G, npix]; {1/e^2, s/e^2} [S, npix]; {sum log e, D.D} [S]; {1/e, s/e} [S, npix];
  // {lam, pix} [G, npix] (the fused objective reads a pixel's two in one request)
  return (int64_t)G * npix + 4ll * S * npix + 2ll * S + 2ll * G * npix;
}
extern "C" int64_t rvs_chisq_work_size(int npix, int S) {
  return rvs_chisq_work_size_g(npix, S, 1);
}

__global__ void __launch_bounds__(256)
    chisq_prepare_kernel(const double *__restrict__ lam,
                         const double *__restrict__ spec,
                         const double *__restrict__ espec, int npix, int S, int G,
                         double x0, double inv_step, int log_step,
                         double espec_sys, double *__restrict__ work) {
  __shared__ double red[8];
  const int s = blockIdx.x;
  double2 *W = reinterpret_cast<double2 *>(work + (int64_t)G * npix);
  double *scal = work + (int64_t)G * npix + 2ll * S * npix;
  double2 *X = reinterpret_cast<double2 *>(scal + 2ll * S);
  if (s >= S) {  // extra blocks: pixel coordinates of grid s - S
    const double lx0 = log(x0);
    const double *lg = lam + (int64_t)(s - S) * npix;
    double *pixa = work + (int64_t)(s - S) * npix;
    double2 *lp = X + (int64_t)S * npix + (int64_t)(s - S) * npix;
    for (int k = threadIdx.x; k < npix; k += 256) {
      const double pv = log_step ? (log(lg[k]) - lx0) * inv_step
                                 : (lg[k] - x0) * inv_step;
      pixa[k] = pv;
      lp[k] = make_double2(lg[k], pv);
    }
    return;
  }
  double lz = 0, dd = 0;
  const double sys2 = espec_sys * espec_sys;
  for (int k = threadIdx.x; k < npix; k += 256) {
    double e = espec[(int64_t)s * npix + k];
    // padding of a short grid (grid sets only): no weight, no term in sum log e.
    // On a single grid an infinite error keeps the reference's arithmetic --
    // log(inf) in the likelihood, a non-finite value the caller is told about
    // (spec_fit.py:963-974)
    if (G > 1 && isinf(e)) {
      W[(int64_t)s * npix + k] = make_double2(0.0, 0.0);
      X[(int64_t)s * npix + k] = make_double2(0.0, 0.0);
      continue;
    }
    if (espec_sys > 0) e = sqrt(sys2 + e * e);
    const double sp = spec[(int64_t)s * npix + k];
    const double d = sp / e;
    lz += log(e);
    dd += d * d;
    W[(int64_t)s * npix + k] = make_double2(1.0 / (e * e), sp / (e * e));
    const double ie = 1.0 / e;
    X[(int64_t)s * npix + k] = make_double2(ie, sp * ie);
  }
  lz = block_sum<4>(lz, red);
  dd = block_sum<4>(dd, red);
  if (threadIdx.x == 0) {
    scal[2 * s] = lz;
    scal[2 * s + 1] = dd;
  }
}

// The G wavelength grids of an arm: spectrum s is on grid gid[s] (nullptr: all on
// grid 0); basis tables of consecutive grids are polys_stride doubles apart.
struct GridSet {
  const int32_t *gid;
  int64_t polys_stride;
  int G;
  const double *pen_scale;   // per-spectrum factor on badchi, or nullptr
};

// packed lower-triangular index
#define TRI(i, j) ((i) * ((i) + 1) / 2 + (j))

// ---------------------------------------------------------------------------
// Wave-uniform operands of the grid kernel's pixel loop, requested ONE PIXEL
// AHEAD of their use.  They are ordinary scalar loads (the compiler tracks what
// is pending and places the s_waitcnt itself); what keeps them -- and the
// per-lane gathers -- at the top of a trip and their first use behind the trip's
// arithmetic is a pair of __builtin_amdgcn_sched_barrier(0) fences (CG_FENCE)
// around the arithmetic and an empty asm that "uses" each value behind it
// (CgRow::pin, cg_use).  An earlier form issued the loads from inline asm, which
// the compiler cannot see through: it measured 2-8 % slower, and under register
// pressure (the resolution-matrix kernel) the allocator split a live range
// right behind such a load -- copied the not-yet-loaded register and reused it
// as an address, which the load then overwrote: a fault.  Nothing here depends
// on instructions the compiler does not know about.
// A basis row of P doubles is held in SGPR tuples that cover it exactly
// (16/8/4/2-dword loads: nothing is read past a row).
// ---------------------------------------------------------------------------
typedef double cg_d8 __attribute__((ext_vector_type(8)));
typedef double cg_d4 __attribute__((ext_vector_type(4)));
typedef double cg_d2 __attribute__((ext_vector_type(2)));
#define CG_FENCE() __builtin_amdgcn_sched_barrier(0)
template <int P>
struct CgRow {
  cg_d8 a0, a1;
  cg_d4 b;
  cg_d2 c;
  double d;
  static constexpr int N8 = P / 8, R = P % 8;
  // request the row at p (rows are 8-byte aligned only)
  __device__ __forceinline__ void load(const double *p) {
    typedef double d8u __attribute__((ext_vector_type(8), aligned(8)));
    typedef double d4u __attribute__((ext_vector_type(4), aligned(8)));
    typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
    if constexpr (N8 >= 1) a0 = *reinterpret_cast<const d8u *>(p);
    if constexpr (N8 >= 2) a1 = *reinterpret_cast<const d8u *>(p + 8);
    if constexpr ((R & 4) != 0) b = *reinterpret_cast<const d4u *>(p + 8 * N8);
    if constexpr ((R & 2) != 0)
      c = *reinterpret_cast<const d2u *>(p + 8 * N8 + (R & 4));
    if constexpr ((R & 1) != 0) d = p[8 * N8 + (R & 6)];
  }
  // first "use" of the row: the loads are waited for here, not earlier
  __device__ __forceinline__ void pin() {
    if constexpr (N8 >= 1) asm volatile("" : "+s"(a0));
    if constexpr (N8 >= 2) asm volatile("" : "+s"(a1));
    if constexpr ((R & 4) != 0) asm volatile("" : "+s"(b));
    if constexpr ((R & 2) != 0) asm volatile("" : "+s"(c));
    if constexpr ((R & 1) != 0) asm volatile("" : "+s"(d));
  }
  // (i is a compile-time constant wherever this is called: unrolled loops)
  __device__ __forceinline__ double get(int i) const {
    if (N8 >= 1 && i < 8) return a0[i];
    if (N8 >= 2 && i < 16) return a1[i - 8];
    int j = i - 8 * N8;
    if ((R & 4) != 0) {
      if (j < 4) return b[j];
      j -= 4;
    }
    if ((R & 2) != 0) {
      if (j < 2) return c[j];
      j -= 2;
    }
    return d;
  }
};
// Structured buffer loads (buffer_load ... idxen): the hardware forms
// base + index * stride from a 32-bit record index, so the per-lane gather of
// knot `pos` (8-B records) and of spline record `pos` (32-B records) needs no
// address arithmetic in the pixel loop.  (The intrinsics have no clang builtin
// in ROCm 7.2; binding the LLVM name is how composable_kernel reaches them.)
typedef int cg_v4i __attribute__((ext_vector_type(4)));
typedef int cg_v2i __attribute__((ext_vector_type(2)));
__device__ cg_v4i cg_sbl128(cg_v4i rsrc, int vindex, int voffset, int soffset,
                            int aux) __asm("llvm.amdgcn.struct.buffer.load.v4i32");
__device__ cg_v2i cg_sbl64(cg_v4i rsrc, int vindex, int voffset, int soffset,
                           int aux) __asm("llvm.amdgcn.struct.buffer.load.v2i32");
__device__ __forceinline__ cg_v4i cg_rsrc(const void *p, unsigned stride,
                                          unsigned nrec) {
  const unsigned long long a = (unsigned long long)p;
  cg_v4i r;
  r.x = (int)(unsigned)a;
  r.y = (int)(((unsigned)(a >> 32) & 0xffffu) | (stride << 16));
  r.z = (int)nrec;
  r.w = 0x00020000;
  return r;
}
// f(integral_constant<int, I>) for I = I0 .. N-1, unrolled at compile time
template <int I, int N, class F>
__device__ __forceinline__ void cg_static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    cg_static_for<I + 1, N>(f);
  }
}
__device__ __forceinline__ void cg_sload2(double &d, const double *p) { d = *p; }
__device__ __forceinline__ void cg_sload4(cg_d2 &d, const double2 *p) {
  typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
  d = *reinterpret_cast<const d2u *>(p);
}

// Measured negative (tools/perf/ubench_dpp.hip, round 2): gfx90a+ allows
// `row_newbcast:n` on 64-bit VALU operations at the full v_fma_f64 rate, so the 65
// uniform factors of a pixel (55 products P_i P_j + 10 P_j) can come from the 16
// lanes of a row of five VGPR pairs instead of SGPRs, which drops the ten
// `P_j * w` multiplications per pixel (74 instead of 84 fp64 instructions).  But
// every lane then loads 5 x 8 B of table per pixel: with the spline gathers that is
// 5 KB per pixel and wave -- 44.2 ms against 34.9 ms per 10 000 spectra.  With
// the table AND the spline window of a 2-wave block staged through LDS by
// global_load_lds (8-pixel chunks, double buffered, no vector-memory instruction
// in the pixel loop, LDS reads in asm so that the compiler does not wait for the
// copy in flight): 74 fp64 + 14 integer instructions per pixel plus the chunk
// bookkeeping -- 35.2-36.4 ms: the ten multiplications saved are spent on LDS
// addressing, and the VALU stays ~80 % busy (now waiting for LDS and the scalar
// cache instead of the gathers).  tools/perf/experiments/chisq_grid_dpp_lds.patch.
// waves per SIMD the register budget is held to (168 / 256 VGPRs)
#define CG_WAVES(P) ((P) <= 10 ? 3 : 2)
// npoly 14..16 (the reference's own tests and the WEAVE driver run 15): the
// P (P + 3) / 2 = 119..152 sums of a lane do not fit the 256 registers of two
// waves per SIMD (round 3: 512 VGPRs + scratch, ONE wave per SIMD).  There the
// pixel loop runs TWICE, the first time for the rows i < CG_SPLIT(P) of the
// normal equations, the second time for the others: a pass holds about half of
// the sums, and every sum still receives its pixels in the same order with the
// same operands -- the values are those of the one-pass loop bit for bit.  What
// it costs is the second evaluation of the spline and of the P products P_j w
// per pixel (215 instead of 175 fp64 operations per pixel-velocity at P = 15).
__host__ __device__ constexpr int cg_split(int P) {
  if (P <= 13) return P;   // (13: 104 sums, 251 VGPRs, no scratch in one pass)
  int pa = 1;
  while (pa < P && pa * (pa + 3) < P * (P + 3) / 2) pa++;
  return pa;
}

// One WAVE per block.  With four waves per block the second block of a
// 400-velocity job (64 + 64 + 16 lanes) held its CU slot as long as a full one:
// 384, 400, 448 and 512 velocities all took the time of 512 (tools/perf/cg_bench).
//
// TAIL = false: the wave owns 64 consecutive velocities of ONE job; everything
// that does not depend on the velocity (basis row, 1/e^2, s/e^2, pixel knot
// coordinate) is wave-uniform and comes through the scalar cache.
// TAIL = true: the Nv % 64 = r left-over velocities of 64/r DIFFERENT jobs share
// a wave (r = 16 for the 400-point grid: four jobs per wave instead of a wave
// with 16 live lanes per job).  Only the spectrum terms {1/e^2, s/e^2} differ
// between the lanes of such a wave; they are fetched per lane (16 B, one
// address per job, base in SGPRs + 32-bit lane offset) -- the basis row is
// still wave-uniform.  A lane's arithmetic is the same sequence of operations
// in both variants, so where a velocity is computed does not change its value.
template <int P, bool TAIL>
__device__ __forceinline__ void
    chisq_grid_body(const int bx, const int by, const double *__restrict__ lam,
                      const double *__restrict__ polysT,
                      const double *__restrict__ work, int npix, int S,
                      const double *__restrict__ knots,
                      const double4 *__restrict__ coef, int ntp, int log_step,
                      const int32_t *__restrict__ job_spec,
                      const int32_t *__restrict__ job_templ, int J,
                      const double *__restrict__ vels, int64_t vel_stride,
                      int Nv, int iv0, int lpj,
                      const double *__restrict__ penalty, double badchi,
                      double beta_out, double *__restrict__ out,
                      int32_t *__restrict__ status, const GridSet GS) {
  // lane -> (job j, velocity index iv)
  int j, iv;
  bool active;
  if (TAIL) {
    // flat map over the J * lpj left-over (job, velocity) pairs: lane g of the
    // launch serves velocity g % lpj of job g / lpj, so every wave but the last
    // is full whatever lpj is (for lpj | 64 this is the 64/lpj-jobs-per-wave
    // layout of round 2, for 100- or 125-point refinement grids -- 36 or 61
    // left over -- it replaces a wave per job with 36 or 61 live lanes)
    const int64_t g = (int64_t)bx * 64 + threadIdx.x;
    j = (int)(g / lpj);
    iv = iv0 + (int)(g - (int64_t)j * lpj);
    active = j < J;
    if (!active) {  // idle lanes shadow the first pair (in-bounds loads)
      j = 0;
      iv = iv0;
    }
  } else {
    j = by;
    iv = bx * 64 + threadIdx.x;
    active = iv < iv0;   // iv0 = number of velocities served by full waves
    if (!active) iv = bx * 64;
  }
  const int s = job_spec ? job_spec[j] : j;
  const int t = job_templ ? job_templ[j] : j;
  double *outp = out + (int64_t)j * Nv;

  const double pen = penalty ? penalty[j] : 0.0;
  // template unusable (non finite outside flag): spec_fit.py:888-893
  const bool unusable = !(pen == pen) || isinf(pen);
  if (GS.pen_scale) badchi *= GS.pen_scale[s];   // 10 x the spectrum's own pixels
  if (!TAIL && unusable) {   // wave-uniform: the whole wave is done
    if (active) {
      const double base = (beta_out != 0.0) ? beta_out * outp[iv] : 0.0;
      outp[iv] = base + 1000.0 * badchi;
    }
    return;
  }

  // the wavelength grid of the spectrum (full waves only: the launcher does not
  // pack lanes of different jobs when the arm has more than one grid)
  // (wave-uniform: the rows below are fetched with scalar loads)
  int gsel = 0;
  if (!TAIL && GS.gid) gsel = GS.gid[__builtin_amdgcn_readfirstlane(s)];
  gsel = __builtin_amdgcn_readfirstlane(gsel);
  lam += (int64_t)gsel * npix;
  polysT += (int64_t)gsel * GS.polys_stride;
  const double *pixa = work + (int64_t)gsel * npix;
  const double2 *W0 =
      reinterpret_cast<const double2 *>(work + (int64_t)GS.G * npix);
  // TAIL: byte offset of the lane's spectrum inside the {1/e^2, s/e^2} block
  // (the launcher checks that it fits 32 bits)
  const uint32_t woff = TAIL ? (uint32_t)s * (uint32_t)npix * 16u : 0u;
  const double2 *W = W0 + (TAIL ? 0 : (int64_t)s * npix);
  const double *scal = work + (int64_t)GS.G * npix + 2ll * S * npix + 2 * s;
  const double4 *cf = coef + (int64_t)t * ntp;

  const double vel = vels[(int64_t)j * vel_stride + iv];
  const double bb = vel / RVS_C_KMS;
  const double f = sqrt((1.0 - bb) / (1.0 + bb));
  const double x0 = knots[0], xlast = knots[ntp - 1];
  double shift;  // knot-coordinate shift of this velocity
  if (log_step) {
    const double inv_step = 1.0 / log(knots[1] / x0);
    shift = log(f) * inv_step;
  } else {
    shift = 0;  // linear knots: coordinate scales instead of shifting
  }
  const double lin_inv_step = log_step ? 0.0 : 1.0 / (knots[1] - x0);

  int32_t st = 0;
  {  // evaler's range test on the first and last pixel (spliner.c:78-83)
    const double xa = lam[0] * f, xb = lam[npix - 1] * f;
    if (xa < x0 || xb < x0 || xa >= xlast || xb >= xlast)
      st |= RVS_ST_SPLINE_RANGE;
  }
  // A grid set pads the spectra on shorter grids up to the longest (weight 0, the
  // grid's last wavelength repeated): the pixel loop stops at the job's OWN last pixel
  // -- a padded pixel adds exact zeros to every sum, so the values are those of the
  // padded loop, and 10 000 SDSS-style spectra of 2842-3842 px no longer pay for 3842
  // each.  The grid's length = the first index that holds its last wavelength
  // (wave-uniform binary search through the scalar cache: ~1 us per wave).
  int npx = npix;
  if (!TAIL && GS.gid) {
    const double last = lam[npix - 1];
    int lo = 0, hi = npix - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (lam[mid] < last)
        lo = mid + 1;
      else
        hi = mid;
    }
    npx = __builtin_amdgcn_readfirstlane(lo + 1);
  }

  double acc[P * (P + 1) / 2];
  double av[P];
#pragma unroll
  for (int i = 0; i < P * (P + 1) / 2; i++) acc[i] = 0;
#pragma unroll
  for (int i = 0; i < P; i++) av[i] = 0;

  // The pixel loop is software-pipelined by hand, one pixel per trip.  While the
  // 75 fp64 operations of pixel k issue, everything pixel k+1 needs is already
  // in flight: its basis row and spectrum terms (scalar loads at the top of the
  // trip), its knot and spline record (per-lane gathers at the top of the trip,
  // from the pixel coordinate fetched one trip earlier).  Left to itself the
  // compiler built a trip as scalar loads -> wait -> positions -> gathers ->
  // wait -> arithmetic: two dependent round trips that the other two waves of
  // the SIMD covered only partly (VALU 87 % busy, waves inside s_waitcnt 39 % of
  // their residency, profiles/r03_sq_counters.json), and every attempt to
  // pipeline it in the source ended with the waits in front of the arithmetic
  // again.  Two scheduling fences around the arithmetic and an
  // empty asm "use" of every prefetched value behind it pin the order; measured
  // inside the kernel (cg_bench -DCG_CLOCK) a SIMD now spends 367 cycles per
  // pixel and wave for 88 VALU instructions x 4 cycles = 352.
  // pos = (int)((log x - log x0)/step) evaluated as pixel coordinate + velocity
  // shift; it can differ from the reference's value only when x is within
  // rounding (~1e-11 knot spacings) of a knot, where the two adjacent cubics agree
  // to O(dx^3) ~ 1e-33 -- exactly the ambiguity the reference's own libm log
  // has (rvs_spline_eval keeps the reference formula verbatim).  The cubic is
  // evaluated in powers of dl = x - x_i (records built with form 1): 3 fma.
  // A lane's arithmetic is the same sequence of operations as in rounds 1-2
  // (P_j * w first, then the fma chain in the order (j, i >= j)).
  auto weights = [&](int k) -> double2 {
    if (TAIL)
      return *reinterpret_cast<const double2 *>(
          reinterpret_cast<const char *>(W0 + k) + woff);
    return W[k];
  };
  // The loop is instantiated per knot spacing (a test inside it became scalar
  // branches per pixel in round 2).
  auto trips = [&](auto log_c, auto i0_c, auto i1_c) {
    constexpr bool LOG = decltype(log_c)::value;
    constexpr int I0 = decltype(i0_c)::value, I1 = decltype(i1_c)::value;
    // clamped knot index of the pixel with wavelength lamk / coordinate pixk
    auto pos_of = [&](double lamk, double pixk, double &x) {
      // rounded product, as numpy's lam * f in the reference: left to itself
      // the compiler contracts x - knot into fma(lam, f, -knot) in the
      // LOG instance (x has no other use there) and chi^2 moves by ~1e-10
      {
#pragma clang fp contract(off)
        x = lamk * f;
      }
      int pos;
      if (LOG)
        pos = (int)(pixk + shift);
      else
        pos = (int)((x - x0) * lin_inv_step);
      int r;
      asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(pos), "s"(ntp - 2));
      return r;
    };
    // full waves: the knot / record gathers as structured buffer loads (the
    // record index is the knot index: no shifts; in round 3's first half, with the
    // loop still waiting for its loads, the same change measured 2 % SLOWER --
    // issue bound, the two instructions saved per pixel are 1.1 %: cg_bench 30.30
    // -> 29.96 ms); packed waves (a template per lane): 32-bit byte offsets
    const cg_v4i rk = cg_rsrc(knots, 8, (unsigned)ntp);
    const cg_v4i rc = cg_rsrc(cf, 32, (unsigned)ntp);
    auto knot_at = [&](int p) -> double {
      if (!TAIL) {
        const cg_v2i v = cg_sbl64(rk, p, 0, 0, 0);
        return __hiloint2double(v.y, v.x);
      }
      return *reinterpret_cast<const double *>(
          reinterpret_cast<const char *>(knots) + ((uint32_t)p << 3));
    };
    auto rec_at = [&](int p) -> double4 {
      if (!TAIL) {
        const cg_v4i lo = cg_sbl128(rc, p, 0, 0, 0);
        const cg_v4i hi = cg_sbl128(rc, p, 16, 0, 0);
        return make_double4(__hiloint2double(lo.y, lo.x),
                            __hiloint2double(lo.w, lo.z),
                            __hiloint2double(hi.y, hi.x),
                            __hiloint2double(hi.w, hi.z));
      }
      return *reinterpret_cast<const double4 *>(
          reinterpret_cast<const char *>(cf) + ((uint32_t)p << 5));
    };
    double w, u;   // (t/e)^2 and t s/e^2 of the pixel whose sums are next
    {
      double x;
      const int p0 = pos_of(lam[0], pixa[0], x);
      const double d0 = x - knot_at(p0);
      const double4 c0 = rec_at(p0);
      const double2 w0 = weights(0);
      const double t0 = fma(fma(fma(c0.w, d0, c0.z), d0, c0.y), d0, c0.x);
      w = t0 * t0 * w0.x;
      u = t0 * w0.y;
    }
    const int klast = npx - 1;
    CgRow<P> R0, R1;
    double la0, pa0, la1, pa1;   // wavelength / knot coordinate of the NEXT pixel
    R0.load(polysT);
    cg_sload2(la0, lam + min(1, klast));
    cg_sload2(pa0, pixa + min(1, klast));
    asm volatile("" : "+s"(la0), "+s"(pa0));
    R0.pin();
    // one pixel: Rc = basis row of pixel k (ready), (lac, pac) = coordinates of
    // pixel k+1 (ready); leaves row k+1 in Rn, coordinates of k+2 in (lan, pan)
    // (the row / coordinate addresses are recomputed from k with scalar
    // instructions; running pointers with an unclamped main loop were measured:
    // no difference, the scalar unit is idle)
    auto trip = [&](int k, CgRow<P> &Rc, CgRow<P> &Rn, double lac, double pac,
                    double &lan, double &pan) {
      const int k1 = min(k + 1, klast), k2 = min(k + 2, klast);
      Rn.load(polysT + (int64_t)k1 * P);
      cg_d2 wn_s;
      if (!TAIL) cg_sload4(wn_s, W + k1);
      cg_sload2(lan, lam + k2);
      cg_sload2(pan, pixa + k2);
      double xn;
      const int pn = pos_of(lac, pac, xn);
      double kn = knot_at(pn);
      double4 cn = rec_at(pn);
      double2 wn_v;
      if (TAIL) wn_v = weights(k1);
      CG_FENCE();
      // the sums of pixel k
#pragma unroll
      for (int jj = 0; jj < I1; jj++) {
        const double pj = Rc.get(jj);
        const double pwj = pj * w;
        if (jj >= I0) av[jj] = fma(pj, u, av[jj]);
#pragma unroll
        for (int i = (jj > I0 ? jj : I0); i < I1; i++)
          acc[TRI(i, jj)] = fma(Rc.get(i), pwj, acc[TRI(i, jj)]);
      }
      CG_FENCE();
      // pixel k+1: its gathers had the whole trip
      asm volatile(""
                   : "+v"(kn), "+v"(cn.x), "+v"(cn.y), "+v"(cn.z), "+v"(cn.w));
      const double dn = xn - kn;
      const double tn = fma(fma(fma(cn.w, dn, cn.z), dn, cn.y), dn, cn.x);
      if (TAIL) {
        asm volatile("" : "+s"(lan), "+s"(pan));
        w = tn * tn * wn_v.x;
        u = tn * wn_v.y;
      } else {
        asm volatile("" : "+s"(lan), "+s"(pan), "+s"(wn_s));
        w = tn * tn * wn_s[0];
        u = tn * wn_s[1];
      }
      Rn.pin();
    };
    int k = 0;
#pragma unroll 1
    for (; k + 1 < npx; k += 2) {
      trip(k, R0, R1, la0, pa0, la1, pa1);
      trip(k + 1, R1, R0, la1, pa1, la0, pa0);
    }
    if (k < npx) trip(k, R0, R1, la0, pa0, la1, pa1);
  };
  constexpr int PS = cg_split(P);
  using c0_t = std::integral_constant<int, 0>;
  using cs_t = std::integral_constant<int, PS>;
  using cp_t = std::integral_constant<int, P>;
  if (log_step) {
    trips(std::true_type{}, c0_t{}, cs_t{});
    if constexpr (PS < P) trips(std::true_type{}, cs_t{}, cp_t{});
  } else {
    trips(std::false_type{}, c0_t{}, cs_t{});
    if constexpr (PS < P) trips(std::false_type{}, cs_t{}, cp_t{});
  }

  // in-lane Cholesky of the packed normal matrix (spec_fit.py:230-247)
  bool ok = true;
  double ldet = 0;
  double pmin = 1.79e308, pmax = 0;   // smallest / largest pivot (squared)
#pragma unroll
  for (int i = 0; i < P; i++) {
#pragma unroll
    for (int jj = 0; jj <= i; jj++) {
      double sum = acc[TRI(i, jj)];
#pragma unroll
      for (int k = 0; k < jj; k++) sum -= acc[TRI(i, k)] * acc[TRI(jj, k)];
      if (jj == i) {
        if (!(sum > 0)) ok = false;
        pmin = fmin(pmin, sum);
        pmax = fmax(pmax, sum);
        const double d = sqrt(sum);
        acc[TRI(i, i)] = d;
        ldet += log(d);
      } else {
        acc[TRI(i, jj)] = sum / acc[TRI(jj, jj)];
      }
    }
  }
  // The orthonormal basis keeps this matrix well conditioned as long as the
  // weights t^2/e^2 are of one order over the arm.  A long stretch of pixels
  // with (nearly) no weight -- half an arm masked with errors inflated 1e6-fold,
  // a template that vanishes over part of the arm -- takes that away: the
  // pivots then span > 1e9 and D.D - y.y loses the 1e-6 of the contract without
  // any pivot turning negative.  Such jobs are flagged; the caller re-evaluates
  // them with rvs_chisq_point (raw basis, explicit residual, in-lane Cholesky:
  // exact to 1e-13 there, tests/test_edge_cases.py) and, where that cannot
  // factor either, with rvs_chisq_full (the eigen tier of spec_fit.py:337-354).
  if (ok && pmin < 1e-9 * pmax) st |= RVS_ST_ILLCOND;
  double yy = 0;
#pragma unroll
  for (int i = 0; i < P; i++) {
    double sum = av[i];
#pragma unroll
    for (int k = 0; k < i; k++) sum -= acc[TRI(i, k)] * av[k];
    av[i] = sum / acc[TRI(i, i)];
    yy = fma(av[i], av[i], yy);
  }
  double chi = 2.0 * ldet + 2.0 * scal[0] + (scal[1] - yy);
  if (st & RVS_ST_SPLINE_RANGE) chi = __builtin_nan("");
  if (!ok) st |= RVS_ST_CHOL_FALLBACK;
  if (!ok || !(fabs(chi) <= 1.79e308)) {
    st |= RVS_ST_NONFINITE;
    chi = __builtin_nan("");
  }
  if (active) {
    const double base = (beta_out != 0.0) ? beta_out * outp[iv] : 0.0;
    if (TAIL && unusable) {
      outp[iv] = base + 1000.0 * badchi;
    } else {
      outp[iv] = base + chi + pen;
      if (st) atomicOr(&status[j], st);
    }
  }
}

#ifdef CG_CLOCK
__device__ unsigned long long cg_clock_dbg[2];
#endif
// Two launches: nfull full waves per job, then the packed waves.  (One launch
// with a block-uniform branch between the two bodies was measured: the full-wave
// path lost 4 % to the shared register allocation, 32.5 against 31.6 ms per
// 10 000 spectra.)
template <int P, bool TAIL>
// (three waves per SIMD are demanded for P <= 10 in both variants: with the
// hand-pipelined loop the compiler has nothing left to reorder, the 168-VGPR
// budget only moves post-loop values to scratch outside the loop.  The packed
// waves had 2 per SIMD at 190 VGPRs in round 2: 100-point grids 9.6 -> 8.9 ms,
// 16-point grids 3.55 -> 2.76 ms per 10 000 jobs with three)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(
    CG_WAVES(P))))
    chisq_grid_kernel(const double *__restrict__ lam,
                      const double *__restrict__ polysT,
                      const double *__restrict__ work, int npix, int S,
                      const double *__restrict__ knots,
                      const double4 *__restrict__ coef, int ntp, int log_step,
                      const int32_t *__restrict__ job_spec,
                      const int32_t *__restrict__ job_templ, int J,
                      const double *__restrict__ vels, int64_t vel_stride,
                      int Nv, int iv0, int lpj, int nfull,
                      const double *__restrict__ penalty, double badchi,
                      double beta_out, double *__restrict__ out,
                      int32_t *__restrict__ status, const GridSet GS) {
  int bx = blockIdx.x, by = 0;
#ifdef CG_CLOCK
  unsigned long long c0 = 0, r0 = 0;
  if (blockIdx.x == 4096) {
    c0 = __builtin_readcyclecounter();
    r0 = wall_clock64();
  }
#endif
  if (!TAIL) {
    // blocks are dealt round-robin over the 8 XCDs (b and b + 8 share one):
    // the nfull waves of a job are 8 blocks apart so that the job's spline
    // records are fetched into ONE L2, once
    // (with one template per CCF node shared by the jobs the placement still
    // pays: cg_bench, 76 node templates: 31.2 against 32.3 ms in job-major order)
    const int per = 8 * nfull;
    const int g = bx / per, r = bx - g * per;
    by = g * 8 + (r & 7);
    bx = r >> 3;
    if (by >= J) return;
  }
  chisq_grid_body<P, TAIL>(bx, by, lam, polysT, work, npix, S,
                           knots, coef, ntp, log_step, job_spec, job_templ, J,
                           vels, vel_stride, Nv, iv0, lpj, penalty, badchi,
                           beta_out, out, status, GS);
#ifdef CG_CLOCK
  if (blockIdx.x == 4096 && threadIdx.x == 0) {
    cg_clock_dbg[0] = __builtin_readcyclecounter() - c0;
    cg_clock_dbg[1] = wall_clock64() - r0;
  }
#endif
}

// ---------------------------------------------------------------------------
// A9: the same kernel with a banded resolution matrix applied to the resampled
// template (convolve_resol, spec_fit.py:474-492, 920-929): model_k =
// sum_d taps[k][d] * raw(k - m + d), m = (nd-1)/2.  A lane owns a velocity and
// walks the pixels in order, so the nd raw values it needs form a sliding
// window: a lane-private ring buffer in LDS (slot = pixel mod nd, [slot][tid]
// layout -> conflict-free), one new spline evaluation per pixel.  The taps of a
// pixel are wave-uniform (scalar loads).  taps [S or 1, npix, nd] row-major,
// taps_stride = npix*nd or 0 when all spectra share one matrix.
// ---------------------------------------------------------------------------
#define RES_MAXND 33

template <int P>
__global__ void __launch_bounds__(256, (P <= 10 ? 2 : 1))
    chisq_grid_resol_kernel(const double *__restrict__ lam,
                            const double *__restrict__ polysT,
                            const double *__restrict__ work, int npix, int S,
                            const double *__restrict__ knots,
                            const double4 *__restrict__ coef, int ntp,
                            int log_step, const double *__restrict__ taps,
                            int nd, int64_t taps_stride,
                            const int32_t *__restrict__ job_spec,
                            const int32_t *__restrict__ job_templ,
                            const double *__restrict__ vels, int64_t vel_stride,
                            int Nv, const double *__restrict__ penalty,
                            double badchi, double beta_out,
                            double *__restrict__ out,
                            int32_t *__restrict__ status, GridSet GS) {
  extern __shared__ double ring[];  // [nd][256]
  const int j = blockIdx.y;
  const int wave_v0 = blockIdx.x * 256 + (threadIdx.x & ~63);
  if (wave_v0 >= Nv) return;
  const int iv = blockIdx.x * 256 + threadIdx.x;
  const bool active = iv < Nv;
  const int s = job_spec ? job_spec[j] : j;
  const int t = job_templ ? job_templ[j] : j;
  double *outp = out + (int64_t)j * Nv;
  const double pen = penalty ? penalty[j] : 0.0;
  if (GS.pen_scale) badchi *= GS.pen_scale[s];
  if (!(pen == pen) || isinf(pen)) {
    if (active) {
      const double base = (beta_out != 0.0) ? beta_out * outp[iv] : 0.0;
      outp[iv] = base + 1000.0 * badchi;
    }
    return;
  }
  // the spectrum's own wavelength grid (grid sets: rvs_chisq_prepare_g's layout)
  const int64_t gi = GS.gid ? GS.gid[s] : 0;
  lam += gi * npix;
  polysT += gi * GS.polys_stride;
  const double *pixa = work + gi * npix;
  const double2 *W = reinterpret_cast<const double2 *>(work + (int64_t)GS.G * npix) +
                     (int64_t)s * npix;
  const double *scal = work + (int64_t)GS.G * npix + 2ll * S * npix + 2 * s;
  const double4 *cf = coef + (int64_t)t * ntp;
  const double *tp = taps + (int64_t)s * taps_stride;
  const int m = (nd - 1) / 2;
  const double vel = vels[(int64_t)j * vel_stride + (active ? iv : 0)];
  const double bb = vel / RVS_C_KMS;
  const double f = sqrt((1.0 - bb) / (1.0 + bb));
  const double x0 = knots[0], xlast = knots[ntp - 1];
  const double shift = log_step ? log(f) / log(knots[1] / x0) : 0.0;
  const double lin_inv_step = log_step ? 0.0 : 1.0 / (knots[1] - x0);
  int32_t st = 0;
  {
    const double xa = lam[0] * f, xb = lam[npix - 1] * f;
    if (xa < x0 || xb < x0 || xa >= xlast || xb >= xlast)
      st |= RVS_ST_SPLINE_RANGE;
  }
  auto raw_at = [&](int p) {
    const double x = lam[p] * f;
    int pos = log_step ? (int)(pixa[p] + shift) : (int)((x - x0) * lin_inv_step);
    pos = min(max(pos, 0), ntp - 2);
    const double dl = x - knots[pos];
    const double4 c = cf[pos];
    return fma(fma(fma(c.w, dl, c.z), dl, c.y), dl, c.x);
  };
  double *mine = ring + threadIdx.x;
  for (int d = 0; d < nd; d++) mine[d * 256] = 0.0;
  for (int p = 0; p < min(m, npix); p++) mine[(p % nd) * 256] = raw_at(p);

  double acc[P * (P + 1) / 2];
  double av[P];
#pragma unroll
  for (int i = 0; i < P * (P + 1) / 2; i++) acc[i] = 0;
#pragma unroll
  for (int i = 0; i < P; i++) av[i] = 0;
  for (int k = 0; k < npix; k++) {
    if (k + m < npix) mine[((k + m) % nd) * 256] = raw_at(k + m);
    const double *tk = tp + (int64_t)k * nd;
    double tv = 0;
    int slot = (k - m + nd) % nd;  // k - m >= -m > -nd
    for (int d = 0; d < nd; d++) {
      tv = fma(tk[d], mine[slot * 256], tv);
      slot = (slot + 1 == nd) ? 0 : slot + 1;
    }
    const double2 wk = W[k];
    const double w = tv * tv * wk.x;
    const double u = tv * wk.y;
    const double *pr = polysT + (int64_t)k * P;
    double pw[P];
#pragma unroll
    for (int i = 0; i < P; i++) pw[i] = pr[i] * w;
#pragma unroll
    for (int i = 0; i < P; i++) {
      av[i] = fma(pr[i], u, av[i]);
#pragma unroll
      for (int jj = 0; jj <= i; jj++)
        acc[TRI(i, jj)] = fma(pr[i], pw[jj], acc[TRI(i, jj)]);
    }
  }
  bool ok = true;
  double ldet = 0;
#pragma unroll
  for (int i = 0; i < P; i++) {
#pragma unroll
    for (int jj = 0; jj <= i; jj++) {
      double sum = acc[TRI(i, jj)];
#pragma unroll
      for (int k = 0; k < jj; k++) sum -= acc[TRI(i, k)] * acc[TRI(jj, k)];
      if (jj == i) {
        if (!(sum > 0)) ok = false;
        const double d = sqrt(sum);
        acc[TRI(i, i)] = d;
        ldet += log(d);
      } else {
        acc[TRI(i, jj)] = sum / acc[TRI(jj, jj)];
      }
    }
  }
  double yy = 0;
#pragma unroll
  for (int i = 0; i < P; i++) {
    double sum = av[i];
#pragma unroll
    for (int k = 0; k < i; k++) sum -= acc[TRI(i, k)] * av[k];
    av[i] = sum / acc[TRI(i, i)];
    yy = fma(av[i], av[i], yy);
  }
  double chi = 2.0 * ldet + 2.0 * scal[0] + (scal[1] - yy);
  if (st & RVS_ST_SPLINE_RANGE) chi = __builtin_nan("");
  if (!ok) st |= RVS_ST_CHOL_FALLBACK;
  if (!ok || !(fabs(chi) <= 1.79e308)) {
    st |= RVS_ST_NONFINITE;
    chi = __builtin_nan("");
  }
  if (active) {
    const double base = (beta_out != 0.0) ? beta_out * outp[iv] : 0.0;
    outp[iv] = base + chi + pen;
    if (st) atomicOr(&status[j], st);
  }
}

// ---- window in registers for a compile-time number of diagonals (11: DESI),
// software-pipelined like chisq_grid_kernel ----
// The window of raw values around the current pixel lives in REGISTERS: the
// pixel loop is unrolled WIN-fold (WIN = ND + 1) so that every slot index
// (pixel mod WIN) is a compile-time constant -- no LDS ring, no moves.
// One wave per block; the wave-uniform operands of pixel k+1 -- basis row,
// spectrum terms, the ND taps of the pixel's resolution row (PER SPECTRUM:
// 88 B a pixel that only this job's seven waves ever read, i.e. a scalar-cache
// miss per pixel) -- are requested at the top of trip k, the knot / record
// gathers of pixel k+M+1 likewise, and their first use stands behind the 75
// fp64 operations of pixel k (CG_FENCE / pin, see CgRow).  The arithmetic
// of a lane is that of the round-2 register-window kernel, operation for
// operation (266 -> 140 ms per step of 10 000 DESI spectra).
template <int P, int ND>
__global__ void __launch_bounds__(64)
    __attribute__((amdgpu_waves_per_eu(P <= 12 ? 2 : 1)))
    chisq_grid_resol_pipe_kernel(const double *__restrict__ lam,
                            const double *__restrict__ polysT,
                            const double *__restrict__ work, int npix, int S,
                            const double *__restrict__ knots,
                            const double4 *__restrict__ coef, int ntp,
                            int log_step, const double *__restrict__ taps,
                            int64_t taps_stride,
                            const int32_t *__restrict__ job_spec,
                            const int32_t *__restrict__ job_templ,
                            const double *__restrict__ vels, int64_t vel_stride,
                            int Nv, const double *__restrict__ penalty,
                            double badchi, double beta_out,
                            double *__restrict__ out,
                            int32_t *__restrict__ status, int J, int nw,
                            GridSet GS) {
  // Blocks are dealt round-robin over the 8 XCDs: the nw waves of a job are 8
  // blocks apart (as in chisq_grid_kernel), so that the job's taps (npix x 11
  // doubles, 242 KB per DESI arm, read by every one of its waves through the
  // scalar cache) and spline records are fetched into ONE L2, once.  With the
  // waves of a job in consecutive blocks -- seven XCDs, seven fetches -- the
  // launch read 44 GB from HBM for 4.8 GB of operands (profiles/r03_pmc_traffic).
  const int per = 8 * nw;
  const int gq = blockIdx.x / per, rq = blockIdx.x - gq * per;
  const int j = gq * 8 + (rq & 7);
  const int bx = rq >> 3;
  if (j >= J) return;
  int iv = bx * 64 + threadIdx.x;
  const bool active = iv < Nv;
  if (!active) iv = bx * 64;
  const int s = job_spec ? job_spec[j] : j;
  const int t = job_templ ? job_templ[j] : j;
  double *outp = out + (int64_t)j * Nv;
  const double pen = penalty ? penalty[j] : 0.0;
  if (GS.pen_scale) badchi *= GS.pen_scale[s];
  if (!(pen == pen) || isinf(pen)) {
    if (active) {
      const double base = (beta_out != 0.0) ? beta_out * outp[iv] : 0.0;
      outp[iv] = base + 1000.0 * badchi;
    }
    return;
  }
  // the spectrum's own wavelength grid (grid sets: rvs_chisq_prepare_g's layout)
  const int64_t gi = GS.gid ? GS.gid[s] : 0;
  lam += gi * npix;
  polysT += gi * GS.polys_stride;
  const double *pixa = work + gi * npix;
  const double2 *W = reinterpret_cast<const double2 *>(work + (int64_t)GS.G * npix) +
                     (int64_t)s * npix;
  const double *scal = work + (int64_t)GS.G * npix + 2ll * S * npix + 2 * s;
  const double4 *cf = coef + (int64_t)t * ntp;
  const double *tp = taps + (int64_t)s * taps_stride;
  const double vel = vels[(int64_t)j * vel_stride + iv];
  const double bb = vel / RVS_C_KMS;
  const double f = sqrt((1.0 - bb) / (1.0 + bb));
  const double x0 = knots[0], xlast = knots[ntp - 1];
  const double shift = log_step ? log(f) / log(knots[1] / x0) : 0.0;
  const double lin_inv_step = log_step ? 0.0 : 1.0 / (knots[1] - x0);
  int32_t st = 0;
  {
    const double xa = lam[0] * f, xb = lam[npix - 1] * f;
    if (xa < x0 || xb < x0 || xa >= xlast || xb >= xlast)
      st |= RVS_ST_SPLINE_RANGE;
  }
  double acc[P * (P + 1) / 2];
  double av[P];
#pragma unroll
  for (int i = 0; i < P * (P + 1) / 2; i++) acc[i] = 0;
#pragma unroll
  for (int i = 0; i < P; i++) av[i] = 0;
  auto pixels = [&](auto log_c) {
    constexpr bool LOG = decltype(log_c)::value;
    constexpr int M = (ND - 1) / 2;
    constexpr int WIN = ND + 1;   // even: the row buffers swap once per pixel
    auto pos_of = [&](double lamk, double pixk, double &x) {
      {  // rounded product, see chisq_grid_kernel
#pragma clang fp contract(off)
        x = lamk * f;
      }
      int pos = LOG ? (int)(pixk + shift) : (int)((x - x0) * lin_inv_step);
      int r;
      asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(pos), "s"(ntp - 2));
      return r;
    };
    // (32-bit byte offsets from the wave-uniform bases; the structured buffer
    // loads of chisq_grid_kernel measured slower here -- two waves per SIMD:
    // 146 against 140 ms per step)
    auto knot_at = [&](int p) {
      return *reinterpret_cast<const double *>(
          reinterpret_cast<const char *>(knots) + ((uint32_t)p << 3));
    };
    auto rec_at = [&](int p) {
      return *reinterpret_cast<const double4 *>(
          reinterpret_cast<const char *>(cf) + ((uint32_t)p << 5));
    };
    const int klast = npix - 1;
    // window of raw values around the current pixel, in registers; slot =
    // pixel mod WIN, a compile-time constant in the WIN-fold unrolled loop
    double win[WIN];
#pragma unroll
    for (int d = 0; d < WIN; d++) win[d] = 0.0;
#pragma unroll
    for (int p = 0; p <= M; p++)
      if (p < npix) {
        double x;
        const int q = pos_of(lam[p], pixa[p], x);
        const double dl = x - knot_at(q);
        const double4 c = rec_at(q);
        win[p % WIN] = fma(fma(fma(c.w, dl, c.z), dl, c.y), dl, c.x);
      }
    double w, u;
    {
      double tv = 0;
#pragma unroll
      for (int d = 0; d < ND; d++)
        tv = fma(tp[d], win[(WIN - M + d) % WIN], tv);
      const double2 w0 = W[0];
      w = tv * tv * w0.x;
      u = tv * w0.y;
    }
    CgRow<P> R0, R1;
    CgRow<ND> TP;
    double la0, pa0, la1, pa1;   // coordinates of pixel k + M + 1 at trip k
    R0.load(polysT);
    cg_sload2(la0, lam + min(M + 1, klast));
    cg_sload2(pa0, pixa + min(M + 1, klast));
    asm volatile("" : "+s"(la0), "+s"(pa0));
    R0.pin();
    auto trip = [&](auto jc, int k, CgRow<P> &Rc, CgRow<P> &Rn, double lac,
                    double pac, double &lan, double &pan) {
      constexpr int JJ = decltype(jc)::value;   // k mod WIN
      const int k1 = min(k + 1, klast), k2 = min(k + M + 2, klast);
      Rn.load(polysT + (int64_t)k1 * P);
      TP.load(tp + (int64_t)k1 * ND);
      cg_d2 wn_s;
      cg_sload4(wn_s, W + k1);
      cg_sload2(lan, lam + k2);
      cg_sload2(pan, pixa + k2);
      const bool fresh = k + M + 1 < npix;   // wave-uniform
      double xn;
      const int pn = pos_of(lac, pac, xn);
      double kn = knot_at(pn);
      double4 cn = rec_at(pn);
      CG_FENCE();
      constexpr int I0 = 0, I1 = P;   // (one pass: the window state is per pixel)
#pragma unroll
      for (int jj = 0; jj < I1; jj++) {
        const double pj = Rc.get(jj);
        const double pwj = pj * w;
        if (jj >= I0) av[jj] = fma(pj, u, av[jj]);
#pragma unroll
        for (int i = (jj > I0 ? jj : I0); i < I1; i++)
          acc[TRI(i, jj)] = fma(Rc.get(i), pwj, acc[TRI(i, jj)]);
      }
      CG_FENCE();
      asm volatile(""
                   : "+v"(kn), "+v"(cn.x), "+v"(cn.y), "+v"(cn.z), "+v"(cn.w));
      const double dn = xn - kn;
      const double raw = fma(fma(fma(cn.w, dn, cn.z), dn, cn.y), dn, cn.x);
      // (beyond the last pixel the matrix has no column: zero, as the sparse
      // product of convolve_resol)
      win[(JJ + M + 1) % WIN] = fresh ? raw : 0.0;
      asm volatile("" : "+s"(lan), "+s"(pan), "+s"(wn_s));
      TP.pin();
      Rn.pin();
      double tv = 0;
#pragma unroll
      for (int d = 0; d < ND; d++)
        tv = fma(TP.get(d), win[(JJ + 1 + WIN - M + d) % WIN], tv);
      w = tv * tv * wn_s[0];
      u = tv * wn_s[1];
    };
    // whole windows without a test per pixel (with one, every trip ended in a
    // branch whose join cost a dozen register moves), then the last < WIN pixels
    int k0 = 0;
#pragma unroll 1
    for (; k0 + WIN <= npix; k0 += WIN) {
      cg_static_for<0, WIN / 2>([&](auto hc) {
        constexpr int J2 = 2 * decltype(hc)::value;
        trip(std::integral_constant<int, J2>{}, k0 + J2, R0, R1, la0, pa0, la1,
             pa1);
        trip(std::integral_constant<int, J2 + 1>{}, k0 + J2 + 1, R1, R0, la1,
             pa1, la0, pa0);
      });
    }
    if (k0 < npix) {
      cg_static_for<0, WIN / 2>([&](auto hc) {
        constexpr int J2 = 2 * decltype(hc)::value;
        if (k0 + J2 < npix)
          trip(std::integral_constant<int, J2>{}, k0 + J2, R0, R1, la0, pa0,
               la1, pa1);
        if (k0 + J2 + 1 < npix)
          trip(std::integral_constant<int, J2 + 1>{}, k0 + J2 + 1, R1, R0, la1,
               pa1, la0, pa0);
      });
    }
  };
  if (log_step)
    pixels(std::true_type{});
  else
    pixels(std::false_type{});
  bool ok = true;
  double ldet = 0;
#pragma unroll
  for (int i = 0; i < P; i++) {
#pragma unroll
    for (int jj = 0; jj <= i; jj++) {
      double sum = acc[TRI(i, jj)];
#pragma unroll
      for (int k = 0; k < jj; k++) sum -= acc[TRI(i, k)] * acc[TRI(jj, k)];
      if (jj == i) {
        if (!(sum > 0)) ok = false;
        const double d = sqrt(sum);
        acc[TRI(i, i)] = d;
        ldet += log(d);
      } else {
        acc[TRI(i, jj)] = sum / acc[TRI(jj, jj)];
      }
    }
  }
  double yy = 0;
#pragma unroll
  for (int i = 0; i < P; i++) {
    double sum = av[i];
#pragma unroll
    for (int k = 0; k < i; k++) sum -= acc[TRI(i, k)] * av[k];
    av[i] = sum / acc[TRI(i, i)];
    yy = fma(av[i], av[i], yy);
  }
  double chi = 2.0 * ldet + 2.0 * scal[0] + (scal[1] - yy);
  if (st & RVS_ST_SPLINE_RANGE) chi = __builtin_nan("");
  if (!ok) st |= RVS_ST_CHOL_FALLBACK;
  if (!ok || !(fabs(chi) <= 1.79e308)) {
    st |= RVS_ST_NONFINITE;
    chi = __builtin_nan("");
  }
  if (active) {
    const double base = (beta_out != 0.0) ? beta_out * outp[iv] : 0.0;
    outp[iv] = base + chi + pen;
    if (st) atomicOr(&status[j], st);
  }
}

// side stream + fork/join events of the calling host thread on the current
// device (host threads drive their own streams: vel_fit.PROCESS_STREAMS)
struct GridFork {
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};
static GridFork *grid_fork() {
  thread_local GridFork fk[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  GridFork &f = fk[dev];
  if (!f.side) {
    if (hipStreamCreateWithFlags(&f.side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&f.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&f.join, hipEventDisableTiming) != hipSuccess) {
      f.side = nullptr;
      (void)hipGetLastError();
      return nullptr;
    }
  }
  return &f;
}

template <int P>
static int launch_grid(const double *lam, const double *polysT,
                       const double *work, int npix, int S, const double *knots,
                       const double *coef, int ntp, int log_step,
                       const int32_t *job_spec, const int32_t *job_templ, int J,
                       const double *vels, int64_t vel_stride, int Nv,
                       const double *penalty, double badchi, double beta,
                       double *out, int32_t *status, int pack_min_jobs,
                       const GridSet GS, hipStream_t st) {
  // left-over velocities of a job (Nv % 64) are packed with those of other jobs
  // in a flat (job, velocity) order, J * r lanes in all.  Thresholds measured
  // with the round-3 kernels (three waves per SIMD in both variants; cg_bench,
  // ms per launch packed / ragged): 10 000 jobs of 100 points (r = 36) 8.8 / 10.0,
  // 112 points (r = 48) 9.8 / 10.2, 125 points (r = 61) 10.7 / 10.2 -> a ragged
  // wave of its own from r = 50 up; 400 points: 1000 jobs 3.7 / 3.7, 2000 jobs
  // 6.6 / 6.9, 4000 jobs 12.2 / 13.3 -> packing from 2000 jobs up (the packed
  // launch alone is one wave time whatever its size).
#ifndef CG_PACK_MAXR
#define CG_PACK_MAXR 50
#endif
  int r = Nv % 64;
  if (pack_min_jobs == 0) pack_min_jobs = 2000;
  if (r > CG_PACK_MAXR || pack_min_jobs < 0 || J < pack_min_jobs ||
      (int64_t)S * npix * 16 >= (1ll << 32) || GS.gid)
    r = 0;   // (grid sets: a packed wave would need a basis row per lane)
  const int nfull = r ? Nv / 64 : (Nv + 63) / 64;   // waves per job, TAIL=false
  const int iv0 = r ? Nv - r : Nv;
  const double4 *cf = reinterpret_cast<const double4 *>(coef);
  // The packed waves go first and on a side stream: alone they are latency bound
  // (2500 waves for 10 000 jobs = 2.4 per SIMD, one dependent scalar-load ->
  // gather round trip per pixel: 4.4 ms per DESI arm for 4 % of the velocities);
  // issued ahead of the full-wave launch they are resident while its waves keep
  // the SIMDs busy (35.2 -> 34.1 ms per arm of 10 000 spectra, same bits).
  GridFork *fk = (r && nfull > 0) ? grid_fork() : nullptr;
  if (r) {
    hipStream_t ts = st;
    if (fk) {
      if (hipEventRecord(fk->fork, st) != hipSuccess ||
          hipStreamWaitEvent(fk->side, fk->fork, 0) != hipSuccess)
        return RVS_E_LAUNCH;
      ts = fk->side;
    }
    hipLaunchKernelGGL((chisq_grid_kernel<P, true>),
                       dim3((unsigned)(((int64_t)J * r + 63) / 64)),
                       dim3(64), 0, ts, lam, polysT, work, npix, S, knots, cf,
                       ntp, log_step, job_spec, job_templ, J, vels, vel_stride,
                       Nv, iv0, r, 0, penalty, badchi, beta, out, status, GS);
    RVS_LAUNCH_CHECK();
    if (fk && hipEventRecord(fk->join, fk->side) != hipSuccess)
      return RVS_E_LAUNCH;
  }
  if (nfull > 0) {
    const int64_t nb = (int64_t)((J + 7) / 8) * 8 * nfull;
    if (nb > 0x7fffffffll) return RVS_E_ARG;
    hipLaunchKernelGGL((chisq_grid_kernel<P, false>), dim3((unsigned)nb),
                       dim3(64), 0, st, lam, polysT, work, npix, S, knots, cf,
                       ntp, log_step, job_spec, job_templ, J, vels, vel_stride,
                       Nv, iv0, 64, nfull, penalty, badchi, beta, out, status,
                       GS);
    RVS_LAUNCH_CHECK();
  }
  if (fk && hipStreamWaitEvent(st, fk->join, 0) != hipSuccess)
    return RVS_E_LAUNCH;
  return 0;
}

extern "C" int rvs_chisq_prepare_g(const double *lam, const double *spec,
                                   const double *espec, int npix, int S, int G,
                                   const double *knots_host3, int log_step,
                                   double espec_sys, double *work, void *stream) {
  // knots_host3: HOST pointer to the first three knots (uniformity test of
  // spliner.c:84-96 is done here, on the host, once per arm)
  if (npix < 1 || S < 1 || G < 1 || !work) return RVS_E_ARG;
  const double x0 = knots_host3[0], x1 = knots_host3[1], x2 = knots_host3[2];
  double step, step2;
  if (log_step) {
    step = log(x1 / x0);
    step2 = log(x2 / x1);
  } else {
    step = x1 - x0;
    step2 = x2 - x1;
  }
  if (fabs(step - step2) > 1e-10) return -3;  // evaler's -2
  hipLaunchKernelGGL(chisq_prepare_kernel, dim3(S + G), dim3(256), 0,
                     rvs_stream(stream), lam, spec, espec, npix, S, G, x0,
                     1.0 / step, log_step, espec_sys, work);
  RVS_LAUNCH_CHECK();
  return 0;
}

extern "C" int rvs_chisq_prepare(const double *lam, const double *spec,
                                 const double *espec, int npix, int S,
                                 const double *knots_host3, int log_step,
                                 double espec_sys, double *work, void *stream) {
  return rvs_chisq_prepare_g(lam, spec, espec, npix, S, 1, knots_host3, log_step,
                             espec_sys, work, stream);
}

extern "C" int rvs_chisq_grid_g(const double *lam, const double *polysT,
                                const double *work, int npix, int npoly, int S,
                                const int32_t *grid_id, int G,
                                int64_t polys_stride, const double *knots,
                                const double *coef, int ntp, int Tn, int log_step,
                                const int32_t *job_spec, const int32_t *job_templ,
                                int J, const double *vels, int64_t vel_stride,
                                int Nv, const double *penalty, double badchi,
                                double beta, int pack_min_jobs,
                                const double *pen_scale, double *out,
                                int32_t *status, void *stream) {
  (void)Tn;
  if (J < 1 || Nv < 1 || npix < 1 || ntp < 3 || G < 1 || (G > 1 && !grid_id))
    return RVS_E_ARG;
  hipStream_t st = rvs_stream(stream);
  const GridSet GS = {G > 1 ? grid_id : nullptr, polys_stride, G, pen_scale};
#define RVS_CASE(PP)                                                          \
  case PP:                                                                    \
    return launch_grid<PP>(lam, polysT, work, npix, S, knots, coef, ntp,      \
                           log_step, job_spec, job_templ, J, vels, vel_stride, \
                           Nv, penalty, badchi, beta, out, status,       \
                           pack_min_jobs, GS, st);
  switch (npoly) {
    RVS_CASE(1) RVS_CASE(2) RVS_CASE(3) RVS_CASE(4) RVS_CASE(5) RVS_CASE(6)
    RVS_CASE(7) RVS_CASE(8) RVS_CASE(9) RVS_CASE(10) RVS_CASE(11) RVS_CASE(12)
    RVS_CASE(13) RVS_CASE(14) RVS_CASE(15) RVS_CASE(16)
    default:
      return RVS_E_ARG;
  }
#undef RVS_CASE
}

extern "C" int rvs_chisq_grid(const double *lam, const double *polysT,
                              const double *work, int npix, int npoly, int S,
                              const double *knots, const double *coef, int ntp,
                              int Tn, int log_step, const int32_t *job_spec,
                              const int32_t *job_templ, int J,
                              const double *vels, int64_t vel_stride, int Nv,
                              const double *penalty, double badchi, double beta,
                              int pack_min_jobs, double *out, int32_t *status,
                              void *stream) {
  return rvs_chisq_grid_g(lam, polysT, work, npix, npoly, S, nullptr, 1, 0, knots,
                          coef, ntp, Tn, log_step, job_spec, job_templ, J, vels,
                          vel_stride, Nv, penalty, badchi, beta, pack_min_jobs,
                          nullptr, out, status, stream);
}

extern "C" int rvs_chisq_grid_resol_g(
    const double *lam, const double *polysT, const double *work, int npix,
    int npoly, int S, const int32_t *grid_id, int G, int64_t polys_stride,
    const double *knots, const double *coef, int ntp, int Tn,
    int log_step, const double *taps, int nd, int64_t taps_stride,
    const int32_t *job_spec, const int32_t *job_templ, int J, const double *vels,
    int64_t vel_stride, int Nv, const double *penalty, double badchi,
    double beta, const double *pen_scale, double *out, int32_t *status,
    void *stream) {
  if (npix < 1 || J < 1 || Nv < 1 || ntp < 3 || Tn < 1 || !taps || nd < 1 ||
      nd > RES_MAXND || (nd & 1) == 0 || J > 65535 || G < 1 || (G > 1 && !grid_id))
    return RVS_E_ARG;
  const GridSet GS = {G > 1 ? grid_id : nullptr, polys_stride, G, pen_scale};
  hipStream_t st = rvs_stream(stream);
  dim3 grid((Nv + 255) / 256, J);
  const size_t shm = (size_t)nd * 256 * sizeof(double);
#define RVS_CASE(PP)                                                           \
  case PP: {                                                                   \
    static bool attr_set = false;                                              \
    if (!attr_set) {                                                           \
      (void)hipFuncSetAttribute((const void *)chisq_grid_resol_kernel<PP>,     \
                                hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                RES_MAXND * 256 * 8);                          \
      (void)hipGetLastError();                                                 \
      attr_set = true;                                                         \
    }                                                                          \
    if (nd == 11)                                                              \
      hipLaunchKernelGGL((chisq_grid_resol_pipe_kernel<PP, 11>),               \
                         dim3((unsigned)(((J + 7) / 8) * 8 * ((Nv + 63) / 64))), \
                         dim3(64), 0, st, lam,                                 \
                         polysT, work, npix, S, knots,                         \
                         reinterpret_cast<const double4 *>(coef), ntp,         \
                         log_step, taps, taps_stride, job_spec, job_templ,     \
                         vels, vel_stride, Nv, penalty, badchi, beta, out,     \
                         status, J, (Nv + 63) / 64, GS);                       \
    else                                                                       \
      hipLaunchKernelGGL(chisq_grid_resol_kernel<PP>, grid, dim3(256), shm,    \
                         st, lam, polysT, work, npix, S, knots,                \
                         reinterpret_cast<const double4 *>(coef), ntp,         \
                         log_step, taps, nd, taps_stride, job_spec, job_templ, \
                         vels, vel_stride, Nv, penalty, badchi, beta, out,     \
                         status, GS);                                          \
  } break;
  switch (npoly) {
    RVS_CASE(1) RVS_CASE(2) RVS_CASE(3) RVS_CASE(4) RVS_CASE(5) RVS_CASE(6)
    RVS_CASE(7) RVS_CASE(8) RVS_CASE(9) RVS_CASE(10) RVS_CASE(11) RVS_CASE(12)
    RVS_CASE(13) RVS_CASE(14) RVS_CASE(15) RVS_CASE(16)
    default:
      return RVS_E_ARG;
  }
#undef RVS_CASE
  RVS_LAUNCH_CHECK();
  return 0;
}

extern "C" int rvs_chisq_grid_resol(
    const double *lam, const double *polysT, const double *work, int npix,
    int npoly, int S, const double *knots, const double *coef, int ntp, int Tn,
    int log_step, const double *taps, int nd, int64_t taps_stride,
    const int32_t *job_spec, const int32_t *job_templ, int J, const double *vels,
    int64_t vel_stride, int Nv, const double *penalty, double badchi,
    double beta, double *out, int32_t *status, void *stream) {
  return rvs_chisq_grid_resol_g(lam, polysT, work, npix, npoly, S, nullptr, 1, 0, knots,
                                coef, ntp, Tn, log_step, taps, nd, taps_stride,
                                job_spec, job_templ, J, vels, vel_stride, Nv, penalty,
                                badchi, beta, nullptr, out, status, stream);
}

// ---------------------------------------------------------------------------
// full output for one velocity per job (spec_fit.py:941-961) and the
// continuum-only fit (spec_fit.py:739-783).  One 256-thread block per job.
// ---------------------------------------------------------------------------
#define FULL_MAXP 32
__device__ void jacobi_eig_dev(double *M, int p, double *w, double *V) {
  for (int i = 0; i < p; i++)
    for (int j = 0; j < p; j++) V[i * p + j] = (i == j);
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    for (int i = 0; i < p; i++)
      for (int j = i + 1; j < p; j++) off += M[i * p + j] * M[i * p + j];
    if (off < 1e-300) break;
    for (int a = 0; a < p; a++)
      for (int b = a + 1; b < p; b++) {
        const double apq = M[a * p + b];
        if (apq == 0) continue;
        const double th = (M[b * p + b] - M[a * p + a]) / (2 * apq);
        const double tt = (th >= 0 ? 1. : -1.) / (fabs(th) + sqrt(th * th + 1));
        const double c = 1 / sqrt(tt * tt + 1), sn = tt * c;
        for (int k = 0; k < p; k++) {
          const double ka = M[k * p + a], kb = M[k * p + b];
          M[k * p + a] = c * ka - sn * kb;
          M[k * p + b] = sn * ka + c * kb;
        }
        for (int k = 0; k < p; k++) {
          const double ak = M[a * p + k], bk = M[b * p + k];
          M[a * p + k] = c * ak - sn * bk;
          M[b * p + k] = sn * ak + c * bk;
        }
        for (int k = 0; k < p; k++) {
          const double ka = V[k * p + a], kb = V[k * p + b];
          V[k * p + a] = c * ka - sn * kb;
          V[k * p + b] = sn * ka + c * kb;
        }
      }
  }
  for (int i = 0; i < p; i++) w[i] = M[i * p + i];
}

__global__ void __launch_bounds__(256)
    chisq_full_kernel(const double *__restrict__ lam,
                      const double *__restrict__ polysT,
                      const double *__restrict__ spec,
                      const double *__restrict__ espec,
                      const uint8_t *__restrict__ badmask, int npix, int P,
                      const double *__restrict__ knots,
                      const double4 *__restrict__ coef, int ntp, int log_step,
                      int cform, int unit_template,
                      const int32_t *__restrict__ job_spec,
                      const int32_t *__restrict__ job_templ,
                      const double *__restrict__ vel, double espec_sys,
                      int fast_interp, const double *__restrict__ taps, int nd,
                      int64_t taps_stride,
                      double *__restrict__ chisq, double *__restrict__ coeffs,
                      double *__restrict__ model, double *__restrict__ raw_model,
                      double *__restrict__ true_chisq, int32_t *__restrict__ ngood,
                      int32_t *__restrict__ status, const GridSet GS) {
  extern __shared__ double sm[];
  double *tvs = sm;                 // [npix]   template / e
  double *Ds = sm + npix;           // [npix]   spec / e
  double *Mm = Ds + npix;           // [P*P]
  double *vv = Mm + FULL_MAXP * FULL_MAXP;  // [P]
  double *aa = vv + FULL_MAXP;      // [P]
  double *red = aa + FULL_MAXP;     // [8]
  double *scr = red + 8;            // [2*P*P + P] jacobi scratch
  __shared__ int sh_st;
  const int j = blockIdx.x;
  const int s = job_spec ? job_spec[j] : j;
  const int t = job_templ ? job_templ[j] : j;
  const int tid = threadIdx.x;
  if (tid == 0) sh_st = 0;
  __syncthreads();
  const double *sp = spec + (int64_t)s * npix;
  const double *es = espec + (int64_t)s * npix;
  if (GS.gid) {   // the spectrum's own wavelength grid and basis
    const int g = GS.gid[s];
    lam += (int64_t)g * npix;
    polysT += (int64_t)g * GS.polys_stride;
  }
  const double sys2 = espec_sys * espec_sys;
  double f = 1, x0 = 0, xlast = 0, inv_step = 0, lx0 = 0;
  const double4 *cf = coef + (int64_t)t * ntp;
  // unit_template == 2: the template ON THE PIXELS, row t of `coef` read as [Tn, npix]
  // doubles (get_chisq0's own argument, spec_fit.py:306-354); no spline, no velocity
  const double *direct = reinterpret_cast<const double *>(coef) + (int64_t)t * npix;
  if (!unit_template) {
    const double bb = vel[j] / RVS_C_KMS;
    f = sqrt((1.0 - bb) / (1.0 + bb));
    x0 = knots[0];
    xlast = knots[ntp - 1];
    inv_step = log_step ? 1.0 / log(knots[1] / x0) : 1.0 / (knots[1] - x0);
    lx0 = log(x0);
    const double xa = lam[0] * f, xb = lam[npix - 1] * f;
    if ((xa < x0 || xb < x0 || xa >= xlast || xb >= xlast) && tid == 0)
      atomicOr(&sh_st, RVS_ST_SPLINE_RANGE);
  }
  double lz = 0;
  if (taps) {
    // A9: raw resampled template (or 1) into Ds, then the banded resolution
    // matrix (convolve_resol, spec_fit.py:474-492): tv_k = sum_d taps[k][d] raw[k-m+d]
    for (int k = tid; k < npix; k += 256) {
      double tv = 1.0;
      if (!unit_template) {
        const double x = lam[k] * f;
        int pos = log_step ? (int)((log(x) - lx0) * inv_step)
                           : (int)((x - x0) * inv_step);
        pos = min(max(pos, 0), ntp - 2);
        const double4 c = cf[pos];
        const double dl = x - knots[pos], dr = knots[pos + 1] - x;
        tv = cform ? fma(fma(fma(c.w, dl, c.z), dl, c.y), dl, c.x)
                   : c.x * dl * dl * dl + c.y * dr * dr * dr + c.z * dl + c.w * dr;
        if (fast_interp)  // nearest knot at or above x (spec_fit.py:913-918)
          tv = cf[min(pos + (dl > 0 ? 1 : 0), ntp - 1)].x;
      } else if (unit_template == 2) {
        tv = direct[k];
      }
      Ds[k] = tv;
    }
    __syncthreads();
    const double *tp = taps + (int64_t)s * taps_stride;
    const int m = (nd - 1) / 2;
    for (int k = tid; k < npix; k += 256) {
      double tv = 0;
      for (int d = 0; d < nd; d++) {
        const int q = k - m + d;
        if (q >= 0 && q < npix) tv = fma(tp[(int64_t)k * nd + d], Ds[q], tv);
      }
      tvs[k] = tv;
    }
    __syncthreads();
  }
  for (int k = tid; k < npix; k += 256) {
    double tv = 1.0;
    if (taps) {
      tv = tvs[k];
    } else if (!unit_template) {
      const double x = lam[k] * f;
      int pos = log_step ? (int)((log(x) - lx0) * inv_step)
                         : (int)((x - x0) * inv_step);
      pos = min(max(pos, 0), ntp - 2);
      const double4 c = cf[pos];
      const double dl = x - knots[pos], dr = knots[pos + 1] - x;
      tv = cform ? fma(fma(fma(c.w, dl, c.z), dl, c.y), dl, c.x)
                 : c.x * dl * dl * dl + c.y * dr * dr * dr + c.z * dl + c.w * dr;
      if (fast_interp)  // nearest knot at or above x (spec_fit.py:913-918)
        tv = cf[min(pos + (dl > 0 ? 1 : 0), ntp - 1)].x;
    } else if (unit_template == 2) {
      tv = direct[k];
    }
    if (raw_model) raw_model[(int64_t)j * npix + k] = tv;
    double e = es[k];
    if (GS.G > 1 && isinf(e)) {   // padding of a short grid (rvs_chisq_prepare_g)
      tvs[k] = 0.0;
      Ds[k] = 0.0;
      continue;
    }
    if (espec_sys > 0) e = sqrt(sys2 + e * e);
    lz += log(e);
    tvs[k] = tv / e;
    Ds[k] = sp[k] / e;
  }
  lz = block_sum<4>(lz, red);
  __syncthreads();
  // normal equations: thread e owns one entry
  const int NE = P * (P + 1) / 2 + P;
  for (int e = tid; e < NE; e += 256) {
    int i, jj;
    bool isv = e >= P * (P + 1) / 2;
    if (isv) {
      i = e - P * (P + 1) / 2;
      jj = 0;
    } else {
      i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
      while (i * (i + 1) / 2 > e) i--;
      while ((i + 1) * (i + 2) / 2 <= e) i++;
      jj = e - i * (i + 1) / 2;
    }
    double sum = 0;
    if (isv) {
      for (int k = 0; k < npix; k++)
        sum += (polysT[(int64_t)k * P + i] * tvs[k]) * Ds[k];
      vv[i] = sum;
    } else {
      for (int k = 0; k < npix; k++)
        sum += (polysT[(int64_t)k * P + i] * tvs[k]) *
               (polysT[(int64_t)k * P + jj] * tvs[k]);
      Mm[i * P + jj] = sum;
      Mm[jj * P + i] = sum;
    }
  }
  __syncthreads();
  __shared__ double sh_ldet;
  if (tid == 0) {
    // Cholesky, eigen fallback (spec_fit.py:337-354)
    double *L = scr;  // P*P
    bool ok = true;
    double ldet = 0;
    for (int i = 0; i < P && ok; i++)
      for (int jj = 0; jj <= i; jj++) {
        double sum = Mm[i * P + jj];
        for (int k = 0; k < jj; k++) sum -= L[i * P + k] * L[jj * P + k];
        if (i == jj) {
          if (!(sum > 0)) {
            ok = false;
            break;
          }
          L[i * P + i] = sqrt(sum);
          ldet += 2 * log(L[i * P + i]);
        } else
          L[i * P + jj] = sum / L[jj * P + jj];
      }
    if (ok) {
      double y[FULL_MAXP];
      for (int i = 0; i < P; i++) {
        double sum = vv[i];
        for (int k = 0; k < i; k++) sum -= L[i * P + k] * y[k];
        y[i] = sum / L[i * P + i];
      }
      for (int i = P - 1; i >= 0; i--) {
        double sum = y[i];
        for (int k = i + 1; k < P; k++) sum -= L[k * P + i] * aa[k];
        aa[i] = sum / L[i * P + i];
      }
      if (!(fabs(ldet) <= 1.79e308)) ok = false;
    }
    if (!ok) {
      atomicOr(&sh_st, RVS_ST_CHOL_FALLBACK);
      double *Wk = scr, *V = scr + P * P;
      double w[FULL_MAXP];
      for (int i = 0; i < P * P; i++) Wk[i] = Mm[i];
      jacobi_eig_dev(Wk, P, w, V);
      ldet = 0;
      for (int i = 0; i < P; i++) ldet += log(fabs(w[i]));
      for (int i = 0; i < P; i++) {
        double sum = 0;
        for (int q = 0; q < P; q++) {
          double vq = 0;
          for (int k = 0; k < P; k++) vq += V[k * P + q] * vv[k];
          sum += V[i * P + q] * vq / w[q];
        }
        aa[i] = sum;
      }
    }
    sh_ldet = ldet;
  }
  __syncthreads();
  double res = 0, tc = 0;
  int ng = 0;
  for (int k = tid; k < npix; k += 256) {
    double m = 0;
    for (int i = 0; i < P; i++) m += aa[i] * polysT[(int64_t)k * P + i];
    const double r = Ds[k] - m * tvs[k];
    res += r * r;
    // model in flux units: coeffs . (polys * templ); true chi^2 uses the
    // ORIGINAL error vector (spec_fit.py:952)
    double e = es[k];
    if (GS.G > 1 && isinf(e)) {
      if (model) model[(int64_t)j * npix + k] = 0.0;
      continue;
    }
    double ee = e;
    if (espec_sys > 0) ee = sqrt(sys2 + e * e);
    const double mod = m * tvs[k] * ee;
    if (model) model[(int64_t)j * npix + k] = mod;
    const double dev = (mod - sp[k]) / e;
    const bool good = badmask ? (badmask[(int64_t)s * npix + k] == 0) : true;
    if (good) {
      tc += dev * dev;
      ng++;
    }
  }
  res = block_sum<4>(res, red);
  tc = block_sum<4>(tc, red);
  const double ngd = block_sum<4>((double)ng, red);
  if (tid == 0) {
    double chi = sh_ldet + 2 * lz + res;
    int st = sh_st;
    if (st & RVS_ST_SPLINE_RANGE) chi = __builtin_nan("");
    if (!(fabs(chi) <= 1.79e308)) st |= RVS_ST_NONFINITE;
    chisq[j] = chi;
    true_chisq[j] = tc;
    ngood[j] = (int)ngd;
    if (st) atomicOr(&status[j], st);
  }
  if (coeffs)
    for (int i = tid; i < P; i += 256) coeffs[(int64_t)j * P + i] = aa[i];
}

extern "C" int rvs_chisq_full_g(const double *lam, const double *polysT,
                              const double *spec, const double *espec,
                              const uint8_t *badmask, int npix, int npoly,
                              int S, const double *knots, const double *coef,
                              int ntp, int Tn, int log_step, int cform,
                              int unit_template, const int32_t *job_spec,
                              const int32_t *job_templ, int J, const double *vel,
                              double espec_sys, int fast_interp,
                              const double *taps, int nd, int64_t taps_stride,
                              double *chisq, double *coeffs, double *model,
                              double *raw_model, double *true_chisq,
                              int32_t *ngood, int32_t *status,
                              const int32_t *grid_id, int G, int64_t polys_stride,
                              void *stream) {
  (void)S;
  if (G < 1 || (G > 1 && !grid_id)) return RVS_E_ARG;
  const GridSet GS = {G > 1 ? grid_id : nullptr, polys_stride, G, nullptr};
  (void)Tn;
  if (npoly < 1 || npoly > FULL_MAXP || J < 1 || npix < 1) return RVS_E_ARG;
  if (taps && (nd < 1 || (nd & 1) == 0)) return RVS_E_ARG;
  if (fast_interp && !cform) return RVS_E_ARG;  // needs y_i in the records
  const size_t shm = sizeof(double) * (2 * (size_t)npix + FULL_MAXP * FULL_MAXP +
                                       2 * FULL_MAXP + 8 +
                                       2 * FULL_MAXP * FULL_MAXP + FULL_MAXP);
  if (shm > 150 * 1024) return RVS_E_ARG;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)chisq_full_kernel,
                        hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipGetLastError();
    attr_set = true;
  }
  hipLaunchKernelGGL(chisq_full_kernel, dim3(J), dim3(256), shm,
                     rvs_stream(stream), lam, polysT, spec, espec, badmask,
                     npix, npoly, knots, reinterpret_cast<const double4 *>(coef),
                     ntp, log_step, cform, unit_template, job_spec, job_templ,
                     vel, espec_sys, fast_interp, taps, nd, taps_stride, chisq,
                     coeffs, model, raw_model, true_chisq, ngood, status, GS);
  RVS_LAUNCH_CHECK();
  return 0;
}

extern "C" int rvs_chisq_full(const double *lam, const double *polysT,
                              const double *spec, const double *espec,
                              const uint8_t *badmask, int npix, int npoly,
                              int S, const double *knots, const double *coef,
                              int ntp, int Tn, int log_step, int cform,
                              int unit_template, const int32_t *job_spec,
                              const int32_t *job_templ, int J, const double *vel,
                              double espec_sys, int fast_interp,
                              const double *taps, int nd, int64_t taps_stride,
                              double *chisq, double *coeffs, double *model,
                              double *raw_model, double *true_chisq,
                              int32_t *ngood, int32_t *status, void *stream) {
  return rvs_chisq_full_g(lam, polysT, spec, espec, badmask, npix, npoly, S, knots,
                          coef, ntp, Tn, log_step, cform, unit_template, job_spec,
                          job_templ, J, vel, espec_sys, fast_interp, taps, nd,
                          taps_stride, chisq, coeffs, model, raw_model, true_chisq,
                          ngood, status, nullptr, 1, 0, stream);
}

// ---------------------------------------------------------------------------
// A13: get_chisq_continuum for a whole batch (spec_fit.py:739-783): template == 1.
// One WAVE per spectrum, lanes = pixels (k = lane, lane + 64, ...): the spectrum
// rows are read coalesced, every lane keeps the P(P+1)/2 + P + 2 sums of its own
// pixels, the wave folds them with DPP (wave_sum_to63) and broadcasts lane 63, all
// lanes factor the P x P matrix, and a second sweep over the pixels forms the model
// residuals of the unmasked pixels.  The order of every sum depends on the pixel
// index only, so a spectrum's result is the same bit for bit alone or among 10 000
// others (tests/test_full_size.py).  One launch.
// (Round 1 had one LANE per spectrum and canonical pixel slices: four launches, every
// load instruction touched 64 cache lines, 1.8 ms per DESI arm of 10 000 spectra.)
__device__ __forceinline__ double wave_total(double v) {
  const double t = wave_sum_to63(v);
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t), 63),
                          __builtin_amdgcn_readlane(__double2loint(t), 63));
}

template <int P>
__global__ void __launch_bounds__(64)
    continuum_wave_kernel(const double *__restrict__ polysT,
                          const double *__restrict__ spec,
                          const double *__restrict__ espec,
                          const double *__restrict__ unit_templ,
                          const uint8_t *__restrict__ badmask, int npix, int S,
                          double *__restrict__ chisq,
                          double *__restrict__ true_chisq,
                          int32_t *__restrict__ ngood,
                          int32_t *__restrict__ status, const GridSet GS) {
  constexpr int NT = P * (P + 1) / 2;
  const int s = blockIdx.x, lane = threadIdx.x;
  if (GS.gid) polysT += (int64_t)GS.gid[s] * GS.polys_stride;
  const double *sp = spec + (int64_t)s * npix;
  const double *es = espec + (int64_t)s * npix;
  const uint8_t *bm = badmask ? badmask + (int64_t)s * npix : nullptr;
  // A9: templ = R @ 1 (spec_fit.py:765-767) instead of 1
  const double *ut = unit_templ ? unit_templ + (int64_t)s * npix : nullptr;
  double acc[NT];
  double av[P];
#pragma unroll
  for (int i = 0; i < NT; i++) acc[i] = 0;
#pragma unroll
  for (int i = 0; i < P; i++) av[i] = 0;
  double lz = 0, dd = 0;
  for (int k = lane; k < npix; k += 64) {
    const double e = es[k], x = sp[k];
    const double ie = 1.0 / e;   // (0 on the padding of a short grid)
    const double tv = ut ? ut[k] : 1.0;
    const double w = (tv * ie) * (tv * ie), u = (tv * ie) * (x * ie);
    lz += (GS.G > 1 && isinf(e)) ? 0.0 : log(e);
    dd = fma(x * ie, x * ie, dd);
    const double *pr = polysT + (int64_t)k * P;
    double prow[P], pw[P];
#pragma unroll
    for (int i = 0; i < P; i++) prow[i] = pr[i];
#pragma unroll
    for (int i = 0; i < P; i++) pw[i] = prow[i] * w;
#pragma unroll
    for (int i = 0; i < P; i++) {
      av[i] = fma(prow[i], u, av[i]);
#pragma unroll
      for (int jj = 0; jj <= i; jj++)
        acc[TRI(i, jj)] = fma(prow[i], pw[jj], acc[TRI(i, jj)]);
    }
  }
#pragma unroll
  for (int i = 0; i < NT; i++) acc[i] = wave_total(acc[i]);
#pragma unroll
  for (int i = 0; i < P; i++) av[i] = wave_total(av[i]);
  lz = wave_total(lz);
  dd = wave_total(dd);
  bool ok = true;
  double ldet = 0;
#pragma unroll
  for (int i = 0; i < P; i++) {
#pragma unroll
    for (int jj = 0; jj <= i; jj++) {
      double sum = acc[TRI(i, jj)];
#pragma unroll
      for (int q = 0; q < jj; q++) sum -= acc[TRI(i, q)] * acc[TRI(jj, q)];
      if (jj == i) {
        if (!(sum > 0)) ok = false;
        const double d = sqrt(sum);
        acc[TRI(i, i)] = d;
        ldet += log(d);
      } else {
        acc[TRI(i, jj)] = sum / acc[TRI(jj, jj)];
      }
    }
  }
  // L y = v ; L^T a = y
#pragma unroll
  for (int i = 0; i < P; i++) {
    double sum = av[i];
#pragma unroll
    for (int q = 0; q < i; q++) sum -= acc[TRI(i, q)] * av[q];
    av[i] = sum / acc[TRI(i, i)];
  }
  double yy = 0;
#pragma unroll
  for (int i = 0; i < P; i++) yy = fma(av[i], av[i], yy);
#pragma unroll
  for (int i = P - 1; i >= 0; i--) {
    double sum = av[i];
#pragma unroll
    for (int q = i + 1; q < P; q++) sum -= acc[TRI(q, i)] * av[q];
    av[i] = sum / acc[TRI(i, i)];
  }
  double tc = 0;
  int ng = 0;
  for (int k = lane; k < npix; k += 64) {
    const double *pr = polysT + (int64_t)k * P;
    double m = 0;
#pragma unroll
    for (int i = 0; i < P; i++) m = fma(av[i], pr[i], m);
    if (ut) m *= ut[k];
    const double dev = (m - sp[k]) / es[k];
    const bool good = bm ? (bm[k] == 0) : true;
    if (good) {
      tc = fma(dev, dev, tc);
      ng++;
    }
  }
  tc = wave_total(tc);
  ng = wave_sum_i(ng);
  if (lane != 0) return;
  int st = 0;
  const double chi = 2.0 * ldet + 2.0 * lz + (dd - yy);
  if (!ok) st |= RVS_ST_CHOL_FALLBACK;
  if (!ok || !(fabs(chi) <= 1.79e308)) st |= RVS_ST_NONFINITE;
  if (chisq) chisq[s] = chi;
  true_chisq[s] = ok ? tc : __builtin_nan("");
  ngood[s] = ng;
  if (st && status) atomicOr(&status[s], st);
}

extern "C" int64_t rvs_chisq_continuum_work_size(int npoly, int S) {
  // (kept in the ABI; the one-wave-per-spectrum kernel needs no scratch: 8 bytes
  // so that callers that allocate what this returns keep working)
  if (npoly < 1 || S < 1) return 0;
  return 8;
}

extern "C" int rvs_chisq_continuum_g(const double *polysT, const double *spec,
                                   const double *espec, const uint8_t *badmask,
                                   const double *unit_templ, int npix, int npoly,
                                   int S, void *work,
                                   double *chisq, double *true_chisq,
                                   int32_t *ngood, int32_t *status,
                                   const int32_t *grid_id, int G,
                                   int64_t polys_stride, void *stream) {
  (void)work;
  if (G < 1 || (G > 1 && !grid_id)) return RVS_E_ARG;
  const GridSet GS = {G > 1 ? grid_id : nullptr, polys_stride, G, nullptr};
  if (S < 1 || npix < 1 || !true_chisq || !ngood) return RVS_E_ARG;
  hipStream_t st = rvs_stream(stream);
#define RVS_CASE(PP)                                                           \
  case PP:                                                                     \
    hipLaunchKernelGGL(continuum_wave_kernel<PP>, dim3(S), dim3(64), 0, st,    \
                       polysT, spec, espec, unit_templ, badmask, npix, S,      \
                       chisq, true_chisq, ngood, status, GS);                  \
    break;
  switch (npoly) {
    RVS_CASE(1) RVS_CASE(2) RVS_CASE(3) RVS_CASE(4) RVS_CASE(5) RVS_CASE(6)
    RVS_CASE(7) RVS_CASE(8) RVS_CASE(9) RVS_CASE(10) RVS_CASE(11) RVS_CASE(12)
    RVS_CASE(13) RVS_CASE(14) RVS_CASE(15) RVS_CASE(16)
    default:
      return RVS_E_ARG;
  }
#undef RVS_CASE
  RVS_LAUNCH_CHECK();
  return 0;
}

extern "C" int rvs_chisq_continuum(const double *polysT, const double *spec,
                                   const double *espec, const uint8_t *badmask,
                                   const double *unit_templ, int npix, int npoly,
                                   int S, void *work,
                                   double *chisq, double *true_chisq,
                                   int32_t *ngood, int32_t *status,
                                   void *stream) {
  return rvs_chisq_continuum_g(polysT, spec, espec, badmask, unit_templ, npix,
                               npoly, S, work, chisq, true_chisq, ngood, status,
                               nullptr, 1, 0, stream);
}

// ---------------------------------------------------------------------------
// A11 at ONE velocity per job (the optimiser's objective, vel_fit.py:205-254):
// one 256-thread BLOCK per (job, arm) -- job = (spectrum idx, own template, own
// velocity) -- threads = pixels, all arms in one launch (grid.y).  Every global
// read is coalesced (spectrum row, basis rows, spline records of neighbouring
// pixels); the P(P+1)/2 + P normal-equation sums are kept per lane, folded
// across each wave with 64-lane butterflies and across the four waves through
// LDS in a fixed order (the velocity-grid kernel needs no reduction because a
// lane owns a velocity; with ONE velocity it would run 1 lane of 64).  Wave 0
// factors the P x P matrix, and a second pass over the pixels forms the
// residual norm ||D - a.ST||^2 explicitly (spec_fit.py:249): no D.D - y.y
// cancellation, so the value can be finite-differenced at any S/N.
// ---------------------------------------------------------------------------
struct PointArms {
  rvs_point_arm a[RVS_MAX_ARMS];
  int n;
};

// the grid of spectrum s on an arm: wavelengths, pixel knot coordinates, basis,
// and the base of the per-spectrum blocks of the rvs_chisq_prepare buffer
struct ArmGrid {
  const double *lam, *pix, *polysT, *wbase;
};
__device__ __forceinline__ ArmGrid arm_grid(const rvs_point_arm &T, int s) {
  ArmGrid g;
  const int G = T.G > 1 ? T.G : 1;
  const int64_t gi = (T.G > 1 && T.grid_id) ? T.grid_id[s] : 0;
  g.lam = T.lam + gi * T.npix;
  g.pix = T.work + gi * T.npix;
  g.polysT = T.polysT + gi * T.polys_stride;
  g.wbase = T.work + (int64_t)G * T.npix;   // {1/e^2, s/e^2} [S, npix], then scal [S, 2]
  return g;
}

__device__ __forceinline__ double point_tv(const rvs_point_arm &T, const ArmGrid &AG,
                                           const double4 *cf, int k, double f,
                                           double shift, double x0,
                                           double lin_inv_step) {
  const double x = AG.lam[k] * f;
  int pos = T.log_step ? (int)(AG.pix[k] + shift)
                       : (int)((x - x0) * lin_inv_step);
  pos = min(max(pos, 0), T.ntp - 2);
  const double dl = x - T.knots[pos];
  if (T.fast_interp) {
    // templ_spec[np.searchsorted(templ_lam, x)] (spec_fit.py:913-918): the
    // first knot >= x; form-1 records carry y_i in .x (also the last row)
    const int idx = min(pos + (dl > 0 ? 1 : 0), T.ntp - 1);
    return cf[idx].x;
  }
  const double4 c = cf[pos];
  return fma(fma(fma(c.w, dl, c.z), dl, c.y), dl, c.x);
}

template <int P>
__global__ void __launch_bounds__(256)
    point_block_kernel(PointArms A, const int32_t *__restrict__ job_spec,
                       const int32_t *__restrict__ job_templ, int J,
                       const double *__restrict__ vel,
                       double *__restrict__ armchi,
                       int32_t *__restrict__ armst) {
  constexpr int NT = P * (P + 1) / 2;
  constexpr int NV = NT + P;
  __shared__ double red[4][NV + 1];
  __shared__ double coefs[P + 2];
  extern __shared__ double rawsh[];  // [npix] only with a resolution matrix
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = blockIdx.x;
  const rvs_point_arm &T = A.a[blockIdx.y];
  const int s = job_spec ? job_spec[j] : j;
  const int t = job_templ ? job_templ[j] : j;
  const ArmGrid AG = arm_grid(T, s);
  const double bb = vel[j] / RVS_C_KMS;
  const double f = sqrt((1.0 - bb) / (1.0 + bb));
  const double espec_sys = T.espec_sys;
  const double sys2 = espec_sys * espec_sys;
  const int npix = T.npix;
  const double *sp = T.spec + (int64_t)s * npix;
  const double *es = T.espec + (int64_t)s * npix;
  const double4 *cf = reinterpret_cast<const double4 *>(T.coef) +
                      (int64_t)t * T.ntp;
  const double x0 = T.knots[0], xlast = T.knots[T.ntp - 1];
  const double shift = T.log_step ? log(f) / log(T.knots[1] / x0) : 0.0;
  const double lin_inv_step = T.log_step ? 0.0 : 1.0 / (T.knots[1] - x0);
  // A9: banded resolution matrix on the resampled template (spec_fit.py:920-929)
  const double *tp = T.taps ? T.taps + (int64_t)s * T.taps_stride : nullptr;
  const int nd = T.nd, mres = (T.nd - 1) / 2;
  if (tp) {
    for (int k = threadIdx.x; k < npix; k += 256)
      rawsh[k] = point_tv(T, AG, cf, k, f, shift, x0, lin_inv_step);
    __syncthreads();
  }
  auto tv_at = [&](int k) {
    if (!tp) return point_tv(T, AG, cf, k, f, shift, x0, lin_inv_step);
    double v = 0;
    for (int d = 0; d < nd; d++) {
      const int q = k - mres + d;
      if (q >= 0 && q < npix) v = fma(tp[(int64_t)k * nd + d], rawsh[q], v);
    }
    return v;
  };
  double acc[NT];
  double av[P];
#pragma unroll
  for (int i = 0; i < NT; i++) acc[i] = 0;
#pragma unroll
  for (int i = 0; i < P; i++) av[i] = 0;
  for (int k = threadIdx.x; k < npix; k += 256) {
    const double tv = tv_at(k);
    double e = es[k];
    if (espec_sys > 0) e = sqrt(sys2 + e * e);
    const double ie = 1.0 / e;
    const double te = tv * ie;
    const double wt = te * te, u = te * (sp[k] * ie);
    const double *pr = AG.polysT + (int64_t)k * P;
    double pv[P], pw[P];
#pragma unroll
    for (int i = 0; i < P; i++) {
      pv[i] = pr[i];
      pw[i] = pv[i] * wt;
    }
#pragma unroll
    for (int i = 0; i < P; i++) {
      av[i] = fma(pv[i], u, av[i]);
#pragma unroll
      for (int jj = 0; jj <= i; jj++)
        acc[TRI(i, jj)] = fma(pv[i], pw[jj], acc[TRI(i, jj)]);
    }
  }
#pragma unroll
  for (int i = 0; i < NT; i++) {
    const double v = wave_sum_to63(acc[i]);
    if (lane == 63) red[w][i] = v;
  }
#pragma unroll
  for (int i = 0; i < P; i++) {
    const double v = wave_sum_to63(av[i]);
    if (lane == 63) red[w][NT + i] = v;
  }
  __syncthreads();
  if (w == 0) {
    // every lane of wave 0 factors the same matrix (waves summed in order)
#pragma unroll
    for (int i = 0; i < NT; i++)
      acc[i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
#pragma unroll
    for (int i = 0; i < P; i++)
      av[i] = ((red[0][NT + i] + red[1][NT + i]) + red[2][NT + i]) +
              red[3][NT + i];
    bool ok = true;
    double ldet = 0;
#pragma unroll
    for (int i = 0; i < P; i++) {
#pragma unroll
      for (int jj = 0; jj <= i; jj++) {
        double sum = acc[TRI(i, jj)];
#pragma unroll
        for (int q = 0; q < jj; q++) sum -= acc[TRI(i, q)] * acc[TRI(jj, q)];
        if (jj == i) {
          if (!(sum > 0)) ok = false;
          const double d = sqrt(sum);
          acc[TRI(i, i)] = d;
          ldet += log(d);
        } else {
          acc[TRI(i, jj)] = sum / acc[TRI(jj, jj)];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < P; i++) {
      double sum = av[i];
#pragma unroll
      for (int q = 0; q < i; q++) sum -= acc[TRI(i, q)] * av[q];
      av[i] = sum / acc[TRI(i, i)];
    }
#pragma unroll
    for (int i = P - 1; i >= 0; i--) {
      double sum = av[i];
#pragma unroll
      for (int q = i + 1; q < P; q++) sum -= acc[TRI(q, i)] * av[q];
      av[i] = sum / acc[TRI(i, i)];
    }
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < P; i++) coefs[i] = av[i];
      coefs[P] = ldet;
      coefs[P + 1] = ok ? 1.0 : 0.0;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < P; i++) av[i] = coefs[i];
  double rr = 0;
  for (int k = threadIdx.x; k < npix; k += 256) {
    const double tv = tv_at(k);
    double e = es[k];
    if (espec_sys > 0) e = sqrt(sys2 + e * e);
    const double ie = 1.0 / e;
    const double *pr = AG.polysT + (int64_t)k * P;
    double m = 0;
#pragma unroll
    for (int i = 0; i < P; i++) m = fma(av[i], pr[i], m);
    const double r = sp[k] * ie - m * (tv * ie);
    rr = fma(r, r, rr);
  }
  rr = wave_sum(rr);
  __syncthreads();
  if (lane == 0) red[w][0] = rr;
  __syncthreads();
  if (threadIdx.x == 0) {
    rr = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
    const double lz = AG.wbase[2ll * T.S * npix + 2 * s];
    double chi = 2.0 * coefs[P] + 2.0 * lz + rr;
    int st = 0;
    const double xa = AG.lam[0] * f, xb = AG.lam[npix - 1] * f;
    if (xa < x0 || xb < x0 || xa >= xlast || xb >= xlast) {
      st |= RVS_ST_SPLINE_RANGE;
      chi = __builtin_nan("");
    }
    const bool ok = coefs[P + 1] != 0.0;
    if (!ok) st |= RVS_ST_CHOL_FALLBACK;
    if (!ok || !(fabs(chi) <= 1.79e308)) {
      st |= RVS_ST_NONFINITE;
      chi = __builtin_nan("");
    }
    armchi[(int64_t)blockIdx.y * J + j] = chi;
    armst[(int64_t)blockIdx.y * J + j] = st;
  }
}

// arms summed in order; penalties of A11 (spec_fit.py:888-896)
__global__ void point_sum_kernel(PointArms A, int J, double badchi,
                                 const int32_t *__restrict__ job_spec,
                                 const double *__restrict__ armchi,
                                 const int32_t *__restrict__ armst,
                                 double *__restrict__ out,
                                 int32_t *__restrict__ status) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= J) return;
  if (A.a[0].pen_scale) badchi *= A.a[0].pen_scale[job_spec ? job_spec[j] : j];
  double tot = 0;
  int st = 0;
  for (int ia = 0; ia < A.n; ia++) {
    const double *pp = A.a[ia].penalty;
    const double pen = pp ? pp[j] : 0.0;
    if (!(pen == pen) || isinf(pen)) {
      tot += 1000.0 * badchi;
      continue;
    }
    tot += armchi[(int64_t)ia * J + j] + pen;
    st |= armst[(int64_t)ia * J + j];
  }
  out[j] = tot;
  if (st) atomicOr(&status[j], st);
}

extern "C" int64_t rvs_chisq_point_work_size(int J, int narm) {
  if (J < 1 || narm < 1) return 0;
  return (int64_t)narm * J * (int64_t)(sizeof(double) + sizeof(int32_t));
}

extern "C" int rvs_chisq_point(const rvs_point_arm *arms, int narm, int npoly,
                               const int32_t *job_spec,
                               const int32_t *job_templ, int J,
                               const double *vel, double badchi, void *scratch,
                               double *out, int32_t *status, void *stream) {
  if (J < 1 || narm < 1 || narm > RVS_MAX_ARMS || !arms || !scratch)
    return RVS_E_ARG;
  PointArms A;
  A.n = narm;
  for (int i = 0; i < narm; i++) {
    A.a[i] = arms[i];
    if (arms[i].npix < 1 || arms[i].ntp < 3) return RVS_E_ARG;
  }
  for (int i = narm; i < RVS_MAX_ARMS; i++) A.a[i] = arms[0];
  hipStream_t st = rvs_stream(stream);
  double *armchi = (double *)scratch;
  int32_t *armst = (int32_t *)(armchi + (int64_t)narm * J);
  dim3 grid(J, narm);
  size_t shm = 0;
  for (int i = 0; i < narm; i++)
    if (arms[i].taps) {
      if (arms[i].nd < 1 || (arms[i].nd & 1) == 0) return RVS_E_ARG;
      shm = max(shm, (size_t)arms[i].npix * sizeof(double));
    }
  if (shm > 60 * 1024) return RVS_E_ARG;
#define RVS_CASE(PP)                                                           \
  case PP:                                                                     \
    hipLaunchKernelGGL(point_block_kernel<PP>, grid, dim3(256), shm, st, A,    \
                       job_spec, job_templ, J, vel, armchi, armst);            \
    break;
  switch (npoly) {
    RVS_CASE(1) RVS_CASE(2) RVS_CASE(3) RVS_CASE(4) RVS_CASE(5) RVS_CASE(6)
    RVS_CASE(7) RVS_CASE(8) RVS_CASE(9) RVS_CASE(10) RVS_CASE(11) RVS_CASE(12)
    RVS_CASE(13) RVS_CASE(14) RVS_CASE(15) RVS_CASE(16)
    default:
      return RVS_E_ARG;
  }
#undef RVS_CASE
  hipLaunchKernelGGL(point_sum_kernel, dim3((J + 255) / 256), dim3(256), 0, st,
                     A, J, badchi, job_spec, armchi, armst, out, status);
  RVS_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
// A12: find_best tail (spec_fit.py:1072-1092).  One 256-thread block / group.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    grid_moments_kernel(const double *__restrict__ chisq,
                        const double *__restrict__ vels, int64_t vel_stride,
                        const int32_t *__restrict__ nvel, int Np, int Nv,
                        int quadratic, double *__restrict__ res,
                        double *__restrict__ probs,
                        int32_t *__restrict__ status) {
  __shared__ double red[8];
  __shared__ double sh_val[4];
  __shared__ long long sh_key[4];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int nv = nvel ? nvel[g] : Nv;
  const double *c = chisq + (int64_t)g * Np * Nv;
  const double *v = vels + (int64_t)g * vel_stride;
  // np.argmin over the reference's [Nv, Np] array: first occurrence in
  // velocity-major order; NaN wins (numpy argmin returns the first NaN)
  double bval = __builtin_inf();
  long long bkey = (1ll << 62);
  bool bnan = false;
  for (int e = tid; e < Np * nv; e += 256) {
    const int p = e / nv, i = e - p * nv;
    const double x = c[(int64_t)p * Nv + i];
    const long long key = (long long)i * Np + p;
    const bool xn = !(x == x);
    bool better;
    if (xn != bnan)
      better = xn;
    else if (xn)
      better = key < bkey;
    else
      better = (x < bval) || (x == bval && key < bkey);
    if (better) {
      bval = x;
      bkey = key;
      bnan = xn;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(bval, o, 64);
    const long long ok = __shfl_xor(bkey, o, 64);
    const bool on = !(ov == ov);
    bool better;
    if (on != bnan)
      better = on;
    else if (on)
      better = ok < bkey;
    else
      better = (ov < bval) || (ov == bval && ok < bkey);
    if (better) {
      bval = ov;
      bkey = ok;
      bnan = on;
    }
  }
  if ((tid & 63) == 0) {
    sh_val[tid >> 6] = bval;
    sh_key[tid >> 6] = bkey;
  }
  __syncthreads();
  bval = sh_val[0];
  bkey = sh_key[0];
  bnan = !(bval == bval);
  for (int w = 1; w < 4; w++) {
    const double ov = sh_val[w];
    const long long ok = sh_key[w];
    const bool on = !(ov == ov);
    bool better;
    if (on != bnan)
      better = on;
    else if (on)
      better = ok < bkey;
    else
      better = (ov < bval) || (ov == bval && ok < bkey);
    if (better) {
      bval = ov;
      bkey = ok;
      bnan = on;
    }
  }
  const int i1 = (int)(bkey / Np), i2 = (int)(bkey % Np);
  const double *col = c + (int64_t)i2 * Nv;
  double psum = 0;
  for (int i = tid; i < nv; i += 256) psum += exp(-0.5 * (col[i] - bval));
  psum = block_sum<4>(psum, red);
  double bv = v[i1];
  int st = 0;
  if (quadratic && i1 > 0 && i1 < nv - 1) {
    // vertex of the parabola through 3 points (np.polyfit deg 2 is exact for
    // 3 points); computed around the centre point to limit cancellation
    const double xa = v[i1 - 1], xb = v[i1], xc = v[i1 + 1];
    const double ya = col[i1 - 1], yb = col[i1], yc = col[i1 + 1];
    const double d1 = (yb - ya) / (xb - xa), d2 = (yc - yb) / (xc - xb);
    const double a2 = (d2 - d1) / (xc - xa);
    const double b1 = d1 + a2 * (xb - xa);  // slope at xb
    bv = xb - b1 / (2 * a2);
    if (!(bv < xc && bv > xa)) st |= RVS_ST_QUAD_ASSERT;
  }
  double m2 = 0, m3 = 0, m4 = 0;
  for (int i = tid; i < nv; i += 256) {
    const double pr = exp(-0.5 * (col[i] - bval)) / psum;
    if (probs) probs[(int64_t)g * Nv + i] = pr;
    const double d = v[i] - bv;
    m2 += pr * d * d;
    m3 += pr * d * d * d;
    m4 += pr * d * d * d * d;
  }
  if (probs)
    for (int i = nv + tid; i < Nv; i += 256) probs[(int64_t)g * Nv + i] = 0;
  m2 = block_sum<4>(m2, red);
  m3 = block_sum<4>(m3, red);
  m4 = block_sum<4>(m4, red);
  if (tid == 0) {
    const double err = sqrt(m2);
    double kur = 0, skw = 0;
    if (!(err < 1e-10)) {
      kur = m4 / (err * err * err * err);
      skw = m3 / (err * err * err);
    }
    double *r = res + (int64_t)g * 8;
    r[0] = bval;
    r[1] = bv;
    r[2] = err;
    r[3] = kur;
    r[4] = skw;
    r[5] = i1;
    r[6] = i2;
    r[7] = psum;
    if (st && status) atomicOr(&status[g], st);
  }
}

extern "C" int rvs_grid_moments(const double *chisq, const double *vels,
                                int64_t vel_stride, const int32_t *nvel, int G,
                                int Np, int Nv, int quadratic, double *res,
                                double *probs, int32_t *status, void *stream) {
  if (G < 1 || Np < 1 || Nv < 1) return RVS_E_ARG;
  hipLaunchKernelGGL(grid_moments_kernel, dim3(G), dim3(256), 0,
                     rvs_stream(stream), chisq, vels, vel_stride, nvel, Np, Nv,
                     quadratic, res, probs, status);
  RVS_LAUNCH_CHECK();
  return 0;
}

extern "C" int rvs_abi_version(void) { return RVS_ABI_VERSION; }
