// ccf.hip -- cross-correlation first guess for gfx950.
//
// Reference: py/rvspecfit/make_ccf.py:105-164, 288-414 (preprocess_data,
// interp_masker, get_continuum, fit_resid) and py/rvspecfit/fitter_ccf.py:62-253.
//
//  ccf_preprocess_kernel  one 512-thread block per spectrum-arm, two blocks per
//      CU: median filter, masks, gap filling, binned-median start, robust
//      (soft-L1) continuum fit as a Levenberg-Marquardt on the same objective,
//      normalisation, 2-point rebin with ivar propagation onto the log-lambda
//      FFT grid.  The medians are exact order statistics found by selection
//      (block-wide histogram rounds, wave-level bisection for the bins).
//  (the FFT cross-correlation itself lives in ccf_fft.hip)
//  ccf_select_kernel      argmin over (template, velocity) + parabola.
#include "common.h"

#define PP_NT 512    // threads per block of the pre-processing kernel (two per CU)
#define PP_LOG2NT 9  // buckets of a selection round = threads
#define PP_NW (PP_NT / 64)


// ---------------------------------------------------------------------------
// Block-wide SELECTION of order statistics (the two plain medians only need the
// middle one or two values; as full bitonic sorts they were 78 barrier-separated
// stages each).  Doubles are mapped to unsigned keys whose integer order is the
// floating-point order (sign flip), the candidates are narrowed by histogram
// rounds -- 1024 buckets over the candidates' key range, prefix sums to find the
// bucket holding rank k: ten bits of the range per round -- until <= 64 remain,
// which one wave ranks directly.  Every step is integer comparison and counting,
// so the value is exactly the one a sort would put at position k.
// ---------------------------------------------------------------------------
#define SEL_NB PP_NT  // buckets == threads: one per thread in the prefix sum
struct SelShared {
  unsigned long long kmin, kmax, result, next, cand[64];
  int hist[SEL_NB];
  int wtot[PP_NW];
  int cnt, nnan, nc, bucket, kk, nle;
};
__device__ __forceinline__ unsigned long long sel_key(double x) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double sel_value(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v,
                                                           int o) {
  const int lo = __shfl_xor((int)(v & 0xffffffffu), o, 64);
  const int hi = __shfl_xor((int)(v >> 32), o, 64);
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}

// numpy median of v[0..n) (LDS); skipnan: np.nanmedian (NaNs are left out),
// else np.median (NaN if there is one).  Called by every thread of the block;
// the value is returned to all of them.
__device__ double block_median(const double *v, int n, bool skipnan,
                               SelShared &Q) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // ---- pass 0: count, NaNs, key range ---------------------------------------
  {
    int cnt = 0, nn = 0;
    unsigned long long mn = ~0ull, mx = 0ull;
    for (int k = tid; k < n; k += PP_NT) {
      const double x = v[k];
      if (x != x) {
        nn++;
        continue;
      }
      const unsigned long long key = sel_key(x);
      cnt++;
      mn = key < mn ? key : mn;
      mx = key > mx ? key : mx;
    }
    cnt = wave_sum_i(cnt);
    nn = wave_sum_i(nn);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long a = shfl_xor_u64(mn, o), b = shfl_xor_u64(mx, o);
      mn = a < mn ? a : mn;
      mx = b > mx ? b : mx;
    }
    if (tid == 0) {
      Q.cnt = 0;
      Q.nnan = 0;
      Q.kmin = ~0ull;
      Q.kmax = 0ull;
    }
    __syncthreads();
    if (lane == 0) {
      atomicAdd(&Q.cnt, cnt);
      atomicAdd(&Q.nnan, nn);
      atomicMin(&Q.kmin, mn);
      atomicMax(&Q.kmax, mx);
    }
    __syncthreads();
  }
  const int nval = Q.cnt;
  if (nval == 0 || (!skipnan && Q.nnan > 0)) {
    __syncthreads();
    return __builtin_nan("");
  }
  const int k_lo = (nval - 1) >> 1, k_hi = nval >> 1;
  unsigned long long lo = Q.kmin, range = Q.kmax - Q.kmin;
  int kk = k_lo, ncand = nval;
  __syncthreads();  // everyone has read Q.kmin / Q.kmax / Q.cnt
  unsigned long long found = 0;
  bool done = false;
  if (range == 0) {
    found = lo;
    done = true;
  }
  while (!done) {
    // bucket = (key - lo) >> shift < SEL_NB
    const int bits = 64 - __clzll((long long)range);  // range >= 1
    const int shift = bits > PP_LOG2NT ? bits - PP_LOG2NT : 0;
    if (ncand <= 64) {
      // ---- the last candidates: ranked by one wave --------------------------
      if (tid == 0) Q.nc = 0;
      __syncthreads();
      for (int k = tid; k < n; k += PP_NT) {
        const double x = v[k];
        if (x != x) continue;
        const unsigned long long key = sel_key(x);
        if (key >= lo && key - lo <= range) Q.cand[atomicAdd(&Q.nc, 1)] = key;
      }
      __syncthreads();
      if (wv == 0) {
        const int nc = Q.nc;
        const unsigned long long my = lane < nc ? Q.cand[lane] : ~0ull;
        int lt = 0, eq = 0;
        for (int j = 0; j < nc; j++) {
          const unsigned long long c = Q.cand[j];
          lt += c < my;
          eq += c == my;
        }
        if (lane < nc && lt <= kk && kk < lt + eq) Q.result = my;
      }
      __syncthreads();
      found = Q.result;
      break;
    }
    Q.hist[tid] = 0;
    __syncthreads();
    for (int k = tid; k < n; k += PP_NT) {
      const double x = v[k];
      if (x != x) continue;
      const unsigned long long key = sel_key(x);
      if (key >= lo && key - lo <= range)
        atomicAdd(&Q.hist[(int)((key - lo) >> shift)], 1);
    }
    __syncthreads();
    // inclusive prefix sum over the buckets (bucket == thread)
    const int h = Q.hist[tid];
    int inc = h;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) Q.wtot[wv] = inc;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wv; w++) off += Q.wtot[w];
    inc += off;
    if (inc - h <= kk && kk < inc) {  // exactly one thread
      Q.bucket = tid;
      Q.kk = kk - (inc - h);
      Q.nc = h;
    }
    __syncthreads();
    const int bsel = Q.bucket;
    kk = Q.kk;
    ncand = Q.nc;
    lo += (unsigned long long)bsel << shift;
    if (shift == 0) {  // buckets of one key
      found = lo;
      break;
    }
    range = (1ull << shift) - 1;
    __syncthreads();  // Q.bucket / Q.kk / Q.nc are rewritten next round
  }
  double med = sel_value(found);
  if (k_hi != k_lo) {
    // the next order statistic: `found` again if it is repeated, else the
    // smallest key above it
    int nle = 0;
    unsigned long long nx = ~0ull;
    for (int k = tid; k < n; k += PP_NT) {
      const double x = v[k];
      if (x != x) continue;
      const unsigned long long key = sel_key(x);
      if (key <= found)
        nle++;
      else
        nx = key < nx ? key : nx;
    }
    nle = wave_sum_i(nle);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long a = shfl_xor_u64(nx, o);
      nx = a < nx ? a : nx;
    }
    __syncthreads();
    if (tid == 0) {
      Q.nle = 0;
      Q.next = ~0ull;
    }
    __syncthreads();
    if (lane == 0) {
      atomicAdd(&Q.nle, nle);
      atomicMin(&Q.next, nx);
    }
    __syncthreads();
    const double hi = (Q.nle >= k_hi + 1) ? med : sel_value(Q.next);
    med = (med + hi) * 0.5;
  }
  __syncthreads();
  return med;
}

// numpy median of up to 256 values held four per lane (x[h] = element lane + 64 h,
// valid while lane + 64 h < cnt) by ONE wave, no LDS: bisection on the key bits
// -- the k-th smallest key is the largest p with #{key < p} <= k, found bit by bit
// with four compares and four ballot counts per bit.  (As a wave-local bitonic
// sort in LDS this was 72 dependent LDS round trips per bin.)  NaN in the bin ->
// NaN, as np.median.
__device__ __forceinline__ double wave_median256(const double (&x)[4], int cnt) {
  const int lane = threadIdx.x & 63;
  unsigned long long key[4];
  bool anynan = false;
#pragma unroll
  for (int h = 0; h < 4; h++) {
    const bool valid = lane + 64 * h < cnt;
    anynan |= valid && (x[h] != x[h]);
    key[h] = valid ? sel_key(x[h]) : ~0ull;
  }
  if (cnt <= 0 || __ballot(anynan) != 0ull) return __builtin_nan("");
  const int k_lo = (cnt - 1) >> 1, k_hi = cnt >> 1;
  // (the padding keys ~0 are never below a trial value, so they never count)
  unsigned long long p = 0;
  for (int bit = 63; bit >= 0; bit--) {
    const unsigned long long trial = p | (1ull << bit);
    int c = 0;
#pragma unroll
    for (int h = 0; h < 4; h++) c += __popcll(__ballot(key[h] < trial));
    if (c <= k_lo) p = trial;
  }
  double med = sel_value(p);
  if (k_hi != k_lo) {
    int cle = 0;
    unsigned long long nx = ~0ull;
#pragma unroll
    for (int h = 0; h < 4; h++) {
      const bool valid = lane + 64 * h < cnt;
      cle += __popcll(__ballot(valid && key[h] <= p));
      if (valid && key[h] > p && key[h] < nx) nx = key[h];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long a = shfl_xor_u64(nx, o);
      nx = a < nx ? a : nx;
    }
    const double hi = (cle >= k_hi + 1) ? med : sel_value(nx);
    med = (med + hi) * 0.5;
  }
  return med;
}

// the 6th smallest of 11 values without NaN: the 28 compare-exchanges of the
// optimal 11-input sorting network (35, 8 layers) that the middle output depends
// on (checked exhaustively with the 0-1 principle); straight-line min / max
__device__ __forceinline__ double median11_net(const double *w) {
  double a[11];
#pragma unroll
  for (int i = 0; i < 11; i++) a[i] = w[i];
#define CE(i, j)                         \
  {                                      \
    const double lo_ = fmin(a[i], a[j]); \
    a[j] = fmax(a[i], a[j]);             \
    a[i] = lo_;                          \
  }
  CE(0, 9) CE(1, 6) CE(2, 4) CE(3, 7) CE(5, 8)
  CE(0, 1) CE(3, 5) CE(4, 10) CE(6, 9) CE(7, 8)
  CE(1, 3) CE(2, 5) CE(4, 7) CE(8, 10)
  CE(0, 4) CE(1, 2) CE(3, 7) CE(5, 9) CE(6, 8)
  CE(2, 6) CE(4, 5) CE(7, 8)
  CE(2, 4) CE(3, 6) CE(5, 7)
  CE(3, 4) CE(5, 6)
  CE(4, 5)
#undef CE
  return a[5];
}

__device__ __forceinline__ double median11(double *a) {
  // insertion sort of 11 values, return the 6th
#pragma unroll
  for (int i = 1; i < 11; i++) {
    const double x = a[i];
    int j = i - 1;
    while (j >= 0 && a[j] > x) {
      a[j + 1] = a[j];
      j--;
    }
    a[j + 1] = x;
  }
  return a[5];
}

#ifdef RVS_PP_TIMING
// debug build only (tools/perf/pp_phases.sh): clock budget of the phases
__device__ unsigned long long pp_dbg[16];
#define PP_T(i)                                                  \
  do {                                                           \
    __syncthreads();                                             \
    if (threadIdx.x == 0) {                                      \
      const unsigned long long t_ = wall_clock64();              \
      atomicAdd(&pp_dbg[i], t_ - t_prev);                        \
      t_prev = t_;                                               \
    }                                                            \
  } while (0)
extern "C" int rvs_dbg_read_pp(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(pp_dbg), sizeof(pp_dbg)) ==
                 hipSuccess
             ? 0
             : -1;
}
#else
#define PP_T(i)
#endif

#define CCF_MAXNODE 24
#ifndef RVS_LM_MAXIT
#define RVS_LM_MAXIT 60
#endif

struct LMShared {
  double p[CCF_MAXNODE];                  // node values (start / result)
  double c[CCF_MAXNODE], cn[CCF_MAXNODE], dl[CCF_MAXNODE];  // B-spline coeffs
  double bvec[CCF_MAXNODE];               // gradient, B-spline space
  double Bd[CCF_MAXNODE * 3];             // banded normal matrix: diag, sub1, sub2
  double Lb[CCF_MAXNODE * 3];             // its LDL^T factor: d, l1, l2
  double isum[CCF_MAXNODE * 9];           // per knot-interval partial sums
  double red[PP_NW];
  double lamd, medv, mederr, medspec;
  int flag, ngood, firstgood, lastgood, nval, stop, ok;
};

// The continuum is exp(clip(S(lam))) with S the k=2 INTERPOLATING spline
// through (nodes, p) (make_ccf.py:155-164).  With C the B-spline collocation
// matrix at the nodes, c = C^-1 p are its B-spline coefficients and
// S(lam_k) = sum_{q<3} Eb[k][q] c[El[k]+q]  (3 non-zero quadratic B-splines per
// pixel).  So every per-pixel quantity costs 3 fma, and the Gauss-Newton
// matrix in node space is H = C^-T (E^T W E) C^-1 with E^T W E pentadiagonal.

// cost 0.5*sum rho(f^2), rho(z) = 2(sqrt(1+z)-1) (scipy soft_l1, f_scale=1) at
// the B-spline coefficients cc.  If store, the model value m_k is left in wm[k]
// (0 where the exponent was clipped): lm_normal forms the Gauss-Newton weights
//   gw = (m/e) f / sqrt(1+z),  hw = (m/e)^2 (1+z)^-1.5
// (rho' f and rho' + 2 rho'' f^2, the scaling scipy applies for robust losses)
// from it.  Storing m instead of (gw, hw) is one LDS array less, which with the
// sort buffer gone lets TWO blocks share a CU (see the kernel).
// px: this thread's first LM_PIX pixels' basis rows and interval indices, read
// once before the iteration (the objective is evaluated ~2x per iteration and
// every evaluation started with two dependent L2 round trips for them); pixels
// beyond LM_PIX * PP_NT fall back to the loads.
struct LMPix {
  double e0, e1, e2;
  int l;
};
// (pixels per thread whose basis rows stay in registers across the iterations:
// at the 128 VGPRs of two blocks per CU more of them spill -- 8.99 / 9.23 /
// 9.60 ms per step for 1 / 3 / 6, tools/perf/pp_variants.sh)
#ifndef LM_PIX
#define LM_PIX 1
#endif

__device__ __forceinline__ double lm_eval(
    LMShared &S, const double *cc, const double *__restrict__ Eb,
    const int32_t *__restrict__ El, int npix, const double *cs, const double *ce,
    double *wm, bool store, const LMPix (&px)[LM_PIX]) {
  __syncthreads();  // cc (LDS) was just written
  double c = 0;
#pragma unroll
  for (int i = 0; i < LM_PIX; i++) {
    const int k = threadIdx.x + i * PP_NT;
    if (k < npix) {
      const int l = px[i].l;
      double s = px[i].e0 * cc[l];
      s = fma(px[i].e1, cc[l + 1], s);
      s = fma(px[i].e2, cc[l + 2], s);
      const bool clipped = (s < -100.0) || (s > 100.0);
      s = fmin(fmax(s, -100.0), 100.0);
      const double mod = exp(s);
      const double e = ce[k];
      const double f = (mod - cs[k]) / e;
      const double z = f * f;
      const double r = sqrt(1 + z);
      c += 2 * (r - 1);
      if (store) wm[k] = clipped ? 0.0 : mod;   // exp(s) > 0: 0 = clipped
    }
  }
  for (int k = threadIdx.x + LM_PIX * PP_NT; k < npix; k += PP_NT) {
    const int l = El[k];
    double s = Eb[3 * k] * cc[l];
    s = fma(Eb[3 * k + 1], cc[l + 1], s);
    s = fma(Eb[3 * k + 2], cc[l + 2], s);
    const bool clipped = (s < -100.0) || (s > 100.0);
    s = fmin(fmax(s, -100.0), 100.0);
    const double mod = exp(s);
    const double e = ce[k];
    const double f = (mod - cs[k]) / e;
    const double z = f * f;
    const double r = sqrt(1 + z);
    c += 2 * (r - 1);
    if (store) wm[k] = clipped ? 0.0 : mod;
  }
  return 0.5 * block_sum<PP_NW>(c, S.red);
}

// Gauss-Newton system in B-SPLINE space: the pentadiagonal E^T W E (S.Bd: diag,
// first and second sub-diagonal) and the gradient E^T gw (S.bvec).
__device__ void lm_normal(LMShared &S, const double *__restrict__ Eb,
                          const int32_t *__restrict__ istart, int m,
                          const double *wm, const double *cs, const double *ce) {
  const int nint = m - 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  for (int t = wave; t < nint; t += PP_NW) {
    double a[9];
#pragma unroll
    for (int q = 0; q < 9; q++) a[q] = 0;
    // four pixels per lane per trip, their basis rows requested together (the
    // rows come out of L2: one exposed round trip per trip instead of per pixel)
    const int k1 = istart[t + 1];
    for (int k0 = istart[t] + lane; k0 < k1; k0 += 4 * 64) {
      double e[4][3];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int k = min(k0 + 64 * c, k1 - 1);
        e[c][0] = Eb[3 * k];
        e[c][1] = Eb[3 * k + 1];
        e[c][2] = Eb[3 * k + 2];
      }
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int k = k0 + 64 * c;
        if (k < k1) {
          const double e0 = e[c][0], e1 = e[c][1], e2 = e[c][2];
          // the weights of lm_eval's formulas, from the stored model value
          const double mod = wm[k], e = ce[k];
          const double f = (mod - cs[k]) / e;
          const double z = f * f;
          const double r = sqrt(1 + z);
          const double d = mod / e;   // (mod = 0: clipped exponent, no weight)
          const double h = d * d / (r * r * r), gg = d * f / r;
          a[0] = fma(h * e0, e0, a[0]);
          a[1] = fma(h * e0, e1, a[1]);
          a[2] = fma(h * e0, e2, a[2]);
          a[3] = fma(h * e1, e1, a[3]);
          a[4] = fma(h * e1, e2, a[4]);
          a[5] = fma(h * e2, e2, a[5]);
          a[6] = fma(gg, e0, a[6]);
          a[7] = fma(gg, e1, a[7]);
          a[8] = fma(gg, e2, a[8]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 9; q++) {
      const double v = wave_sum_to63(a[q]);  // DPP on the VALU, total in lane 63
      if (lane == 63) S.isum[t * 9 + q] = v;
    }
  }
  __syncthreads();
  // banded assembly: entry (a, a-d), d = 0,1,2 ; interval t covers coefficients
  // t..t+2 and holds {00,01,02,11,12,22 | g0,g1,g2}
  for (int e = threadIdx.x; e < 4 * m; e += PP_NT) {
    const int a = e >> 2, d = e & 3;
    if (d == 3) {
      double s = 0;
      for (int t = max(0, a - 2); t <= min(a, nint - 1); t++)
        s += S.isum[t * 9 + 6 + (a - t)];
      S.bvec[a] = s;
    } else {
      const int lo = a - d;
      double s = 0;
      if (lo >= 0) {
        for (int t = max(0, a - 2); t <= min(lo, nint - 1); t++) {
          const int u = lo - t, v = a - t;  // 0<=u<=v<=2
          const int idx = (u == 0) ? v : (u == 1 ? 2 + v : 5);
          s += S.isum[t * 9 + idx];
        }
      }
      S.Bd[a * 3 + d] = s;
    }
  }
  __syncthreads();
}

// (B + lamd diag B) dl = -bvec with B pentadiagonal SPD: LDL^T with a sliding
// window, O(m).  Levenberg-Marquardt damping in B-spline space.
// Run by ONE WAVE (all lanes execute the same chain).  Row i of the banded
// system lives in lane i's registers; the serial chain fetches it with
// v_readlane and parks its results (l1_i, l2_i, w_i) in lane i (a select:
// all lanes hold the same value), so neither sweep touches LDS (as a single-lane loop over LDS
// every row cost four dependent LDS round trips and three divisions; here one
// division, 1 / d_i, serves the three quotients by d_i, d_{i-1}, d_{i-2}).
__device__ inline double lane_get(double v, int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src),
                          __builtin_amdgcn_readlane(__double2loint(v), src));
}
// every lane holds the same v: lane `lane` keeps it
__device__ inline double lane_put(double dst, double v, int lane) {
  return ((int)(threadIdx.x & 63) == lane) ? v : dst;
}

__device__ void lm_band_solve(LMShared &S, int m) {
  const int lane = threadIdx.x & 63;
  const int r = min(lane, m - 1);
  const double ld = 1.0 + S.lamd;
  // this lane's row: diagonal (damped), first and second sub-diagonal, gradient
  const double ra0 = S.Bd[r * 3] * ld, ra1 = S.Bd[r * 3 + 1],
               ra2 = S.Bd[r * 3 + 2], rb = S.bvec[r];
  double d1 = 0, d2 = 0, l1p = 0;  // d_{i-1}, d_{i-2}, l1_{i-1}
  double r1 = 0, r2 = 0;           // 1 / d_{i-1}, 1 / d_{i-2}
  double z1 = 0, z2 = 0;           // z_{i-1}, z_{i-2}
  double lb1 = 0, lb2 = 0, w = 0;  // row results, one row per lane
  bool ok = true;
  for (int i = 0; i < m; i++) {
    const double a0 = lane_get(ra0, i), a1 = lane_get(ra1, i),
                 a2 = lane_get(ra2, i), bi = lane_get(rb, i);
    const double l2 = (i >= 2) ? a2 * r2 : 0.0;
    const double l1 = (i >= 1) ? (a1 - l2 * l1p * d2) * r1 : 0.0;
    const double d = a0 - l1 * l1 * d1 - l2 * l2 * d2;
    if (!(d > 0)) ok = false;
    const double z = -bi - l1 * z1 - l2 * z2;
    const double rd = 1.0 / d;
    lb1 = lane_put(lb1, l1, i);
    lb2 = lane_put(lb2, l2, i);
    w = lane_put(w, z * rd, i);  // w_i = z_i / d_i
    d2 = d1;
    d1 = d;
    r2 = r1;
    r1 = rd;
    l1p = l1;
    z2 = z1;
    z1 = z;
  }
  double x1 = 0, x2 = 0;  // x_{i+1}, x_{i+2}
  double xs = 0;          // solution, one component per lane
  for (int i = m - 1; i >= 0; i--) {
    const double l1n = (i + 1 < m) ? lane_get(lb1, i + 1) : 0.0;
    const double l2n = (i + 2 < m) ? lane_get(lb2, i + 2) : 0.0;
    const double x = lane_get(w, i) - l1n * x1 - l2n * x2;
    xs = lane_put(xs, x, i);
    x2 = x1;
    x1 = x;
  }
  if (lane < m) {
    const double dl = ok ? xs : 0.0;
    S.dl[lane] = dl;
    S.cn[lane] = S.c[lane] + dl;
  }
  if (lane == 0) S.ok = ok ? 1 : 0;
}

__global__ void __launch_bounds__(PP_NT)
    __attribute__((amdgpu_waves_per_eu(4, 4)))  // two 8-wave blocks per CU
    ccf_preprocess_kernel(const double *__restrict__ lam,
                          const double *__restrict__ spec,
                          const double *__restrict__ espec,
                          const uint8_t *__restrict__ badmask, int npix,
                          int continuum, const double *__restrict__ Eb,
                          const int32_t *__restrict__ El,
                          const double *__restrict__ Cinv,
                          const int32_t *__restrict__ istart, int nnode,
                          const int32_t *__restrict__ bin_start,
                          const int32_t *__restrict__ xind,
                          const double *__restrict__ rw, int nfft, double maxerr,
                          double *__restrict__ proc_spec,
                          double *__restrict__ proc_ivar, double *__restrict__ sse,
                          double *__restrict__ cont_out,
                          double *__restrict__ pfit, int32_t *__restrict__ status,
                          const int32_t *__restrict__ gid,
                          const int32_t *__restrict__ npix_g,
                          const int32_t *__restrict__ nnode_g) {
  extern __shared__ double sm[];
  // Grid sets (rvs_ccf_preprocess_g): the spectrum of this block is observed on
  // grid g = gid[b] with npix_g[g] pixels and nnode_g[g] continuum nodes; every
  // table holds one slice per grid, npix / nnode (the kernel arguments) apart.
  const int npix_s = npix, nnode_s = nnode;   // row strides (= the largest grid)
  if (gid) {
    const int g = gid[blockIdx.x];
    npix = npix_g[g];
    nnode = nnode_g[g];
    lam += (int64_t)g * npix_s;
    xind += (int64_t)g * nfft;
    rw += (int64_t)g * nfft;
    if (continuum) {
      Eb += (int64_t)g * 3 * npix_s;
      El += (int64_t)g * npix_s;
      Cinv += (int64_t)g * 2 * nnode_s * nnode_s;
      istart += (int64_t)g * nnode_s;
      bin_start += (int64_t)g * (nnode_s + 1);
    }
  }
  // Three arrays of npix doubles + the mask: 72 KB for a DESI arm, so that TWO
  // 512-thread blocks share a CU.  The kernel is a chain of short phases that
  // barriers and single-wave steps (the LM band solve) separate: with ONE
  // 1024-thread block per CU (round 2: 110 KB with the (gw, hw) pair and a
  // 4096-entry sort buffer for bins of more than 256 pixels) the VALU was 45 %
  // busy, and halving the threads of that lone block cost 6 % -- the time is in
  // the chain, which a second resident block covers.
  double *cs = sm;              // [npix] current spectrum
  double *ce = cs + npix;       // [npix] current error
  double *wm = ce + npix;       // [npix] model values of the last evaluation
  uint8_t *msk = reinterpret_cast<uint8_t *>(wm + npix);  // [npix]
  __shared__ LMShared S;
  __shared__ SelShared Q;
  const int b = blockIdx.x, tid = threadIdx.x;
  const double *sp0 = spec + (int64_t)b * npix_s;
  const double *es0 = espec + (int64_t)b * npix_s;
  const double nanv = __builtin_nan("");
#ifdef RVS_PP_TIMING
  unsigned long long t_prev = wall_clock64();
#endif

  for (int k = tid; k < npix; k += PP_NT) {
    cs[k] = sp0[k];
    ce[k] = es0[k];
    msk[k] = badmask ? (badmask[(int64_t)b * npix_s + k] != 0) : 0;
  }
  if (tid == 0) {
    S.flag = 0;
    S.nval = 0;
  }
  __syncthreads();

  // ---- nanmedian(espec) -----------------------------------------------------
  {
    const double me = block_median(ce, npix, true, Q);
    if (tid == 0) S.mederr = me;
    __syncthreads();
  }
  const double mederr = S.mederr;

  // ---- medfilt(spec, 11) <= 0 and error clipping (make_ccf.py:366-370) -------
  if (continuum) {
    for (int k = tid; k < npix; k += PP_NT) {
      double w[11];
#pragma unroll
      for (int q = 0; q < 11; q++) {
        const int idx = k + q - 5;
        w[q] = (idx >= 0 && idx < npix) ? cs[idx] : 0.0;  // zero padded
      }
      // (a window that holds a NaN keeps the comparison order of the insertion
      // sort; everywhere else the middle value does not depend on the method)
      bool hasnan = false;
#pragma unroll
      for (int q = 0; q < 11; q++) hasnan |= (w[q] != w[q]);
      const double med = hasnan ? median11(w) : median11_net(w);
      if ((ce[k] > maxerr * mederr) || (med <= 0)) msk[k] = 1;
    }
  }
  __syncthreads();
  PP_T(0);  // load + nanmedian(espec) sort + medfilt + masks
  // ---- inflate masked errors, fill gaps (interp_masker, make_ccf.py:288-327) --
  {
    int ng = 0, fg = npix, lg = -1;
    for (int k = tid; k < npix; k += PP_NT) {
      if (msk[k]) {
        ce[k] = 1e9 * mederr;
      } else {
        ng++;
        fg = min(fg, k);
        lg = max(lg, k);
      }
    }
    ng = wave_sum_i(ng);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      fg = min(fg, __shfl_xor(fg, o, 64));
      lg = max(lg, __shfl_xor(lg, o, 64));
    }
    if (tid == 0) {
      S.ngood = 0;
      S.firstgood = npix;
      S.lastgood = -1;
    }
    __syncthreads();
    if ((tid & 63) == 0) {
      atomicAdd(&S.ngood, ng);
      atomicMin(&S.firstgood, fg);
      atomicMax(&S.lastgood, lg);
    }
    __syncthreads();
    if (S.ngood == 0) {
      if (tid == 0) S.flag |= RVS_ST_ALLMASKED;
      for (int k = tid; k < npix; k += PP_NT)
        if (!(fabs(cs[k]) <= 1.79e308)) cs[k] = 1;
    } else {
      const int fgood = S.firstgood, lgood = S.lastgood;
      for (int k = tid; k < npix; k += PP_NT) {
        if (!msk[k]) continue;
        double val;
        if (k < fgood)
          val = sp0[fgood];
        else if (k > lgood)
          val = sp0[lgood];
        else {
          int a = k - 1, c = k + 1;
          while (msk[a]) a--;
          while (msk[c]) c++;
          const double l1 = lam[a], l2 = lam[c], l0 = lam[k];
          val = (-(l1 - l0) * sp0[c] + (l2 - l0) * sp0[a]) / (l2 - l1);
        }
        cs[k] = val;  // only masked pixels are written, only good ones read
      }
    }
    __syncthreads();
  }

  PP_T(1);  // gap filling
  // ---- median of the filled spectrum ----------------------------------------
  {
    const double md = block_median(cs, npix, false, Q);
    if (tid == 0) {
      S.medv = md;
      double ms = md;
      if (ms <= 0) {
        ms = fabs(ms);
        if (ms == 0) ms = 1;
      }
      S.medspec = ms;
    }
    __syncthreads();
  }
  const double medv = S.medv;

  if (continuum) {
    const int m = nnode;
    PP_T(2);  // median sort
    // ---- binned medians -> p0 (make_ccf.py:141-143) -------------------------
    // bins of <= 256 pixels (the usual case: ~140): a wave takes a whole bin, four
    // values per lane in registers, 8 bins at a time, no block barriers;
    // otherwise the block-wide selection, bin after bin
    bool small = true;
    for (int jb = 0; jb < m; jb++)
      if (bin_start[jb + 1] - bin_start[jb] > 256) small = false;
    if (small) {
      const int lane = tid & 63, wv = tid >> 6;
      for (int jb = wv; jb < m; jb += PP_NW) {
        const int b0 = bin_start[jb], cnt = bin_start[jb + 1] - b0;
        double x[4];
#pragma unroll
        for (int h = 0; h < 4; h++) {
          const int q = lane + 64 * h;
          x[h] = (q < cnt) ? cs[b0 + q] : 0.0;
        }
        const double stat = wave_median256(x, cnt);
        if (lane == 0) {
          double p0 = log(fmax(stat, 1e-3 * S.medspec));
          if (!(stat == stat)) p0 = nanv;  // np.maximum propagates NaN
          if (!(fabs(p0) <= 1.79e308)) p0 = log(S.medspec);
          S.p[jb] = p0;
        }
      }
    } else {
      // a bin of more than 256 pixels: np.median of each bin by the block-wide
      // selection (block_median: exact order statistics, NaN if the bin holds
      // one), bin after bin
      for (int jb = 0; jb < m; jb++) {
        const int b0 = bin_start[jb], cnt = bin_start[jb + 1] - b0;
        const double stat = block_median(cs + b0, cnt, false, Q);
        if (tid == 0) {
          double p0 = log(fmax(stat, 1e-3 * S.medspec));
          if (!(stat == stat)) p0 = nanv;  // np.maximum propagates NaN
          if (!(fabs(p0) <= 1.79e308)) p0 = log(S.medspec);
          S.p[jb] = p0;
        }
      }
    }
    __syncthreads();

    PP_T(3);  // binned-median sort
    // ---- Levenberg-Marquardt on the soft-L1 objective -----------------------
    // unknowns: the B-spline coefficients c = C^-1 p of the interpolating spline
    // (a linear bijection of the reference's node values p, same minimum)
    if (tid < m) {
      double s = 0;
      for (int jj = 0; jj < m; jj++) s = fma(Cinv[tid * m + jj], S.p[jj], s);
      S.c[tid] = s;
    }
    if (tid == 0) {
#ifndef RVS_LM_LAMBDA0
#define RVS_LM_LAMBDA0 1e-3
#endif
      S.lamd = RVS_LM_LAMBDA0;
      S.stop = 0;
    }
    LMPix px[LM_PIX];
#pragma unroll
    for (int i = 0; i < LM_PIX; i++) {
      const int k = min(tid + i * PP_NT, npix - 1);
      px[i].e0 = Eb[3 * k];
      px[i].e1 = Eb[3 * k + 1];
      px[i].e2 = Eb[3 * k + 2];
      px[i].l = El[k];
    }
    double cost = lm_eval(S, S.c, Eb, El, npix, cs, ce, wm, true, px);
    PP_T(7);  // (debug) LM set-up + first evaluation
    for (int it = 0; it < RVS_LM_MAXIT; it++) {
      lm_normal(S, Eb, istart, m, wm, cs, ce);
      PP_T(8);  // (debug) normal equations
      if (it == 0) {
        // least_squares' gtol exit at the starting point (make_ccf.py:146-150,
        // gtol = 1e-8 absolute): with |J^T rho' f|_inf below it TRF returns p0
        // untouched -- what happens when the errors dwarf the flux.  The
        // gradient with respect to the node values p is C^-T (E^T gw).
        if (tid < 64) {
          double g = 0;
          if (tid < m)
            for (int i = 0; i < m; i++) g = fma(Cinv[i * m + tid], S.bvec[i], g);
          g = fabs(g);
          if (!(g == g)) g = 1.0;  // NaN gradient: not an exit
          for (int o = 32; o > 0; o >>= 1) g = fmax(g, __shfl_xor(g, o));
          if (tid == 0 && g < 1e-8) S.stop = 1;
        }
        __syncthreads();
        if (S.stop) break;
      }
      // damped step; retry with larger damping until the cost does not grow.
      // The trial evaluation leaves its model values in wm: they are
      // read by the NEXT iteration's lm_normal only (a retry re-solves the banded
      // system in S), so an accepted trial needs no second evaluation -- one
      // pass over the pixels less per iteration, same values.
      bool accepted = false;
      for (int tries = 0; tries < 40; tries++) {
        if (tid < 64) lm_band_solve(S, m);
        PP_T(9);  // (debug) band solve
        const double cn =
            lm_eval(S, S.cn, Eb, El, npix, cs, ce, wm, true, px);
        PP_T(10);  // (debug) trial evaluation
        if (cn <= cost) {  // accept (block-uniform decision)
          accepted = true;
          double mx = 0;
          for (int i = 0; i < m; i++) mx = fmax(mx, fabs(S.dl[i]));
          const double rel = (cost - cn) / fmax(cost, 1e-300);
          __syncthreads();
          if (tid == 0) {
            for (int i = 0; i < m; i++) S.c[i] = S.cn[i];
            S.lamd = fmax(S.lamd / 8, 1e-12);
            // converged: the accepted step moved no coefficient (log flux) by
            // 1e-9 or lowered the cost by less than 1e-12 of itself -- four orders
            // inside least_squares' own ftol = xtol = 1e-8 (make_ccf.py:146-150)
            if (mx < 1e-9 || rel < 1e-12) S.stop = 1;
          }
          cost = cn;
          break;
        }
        __syncthreads();
        if (tid == 0) {
          S.lamd *= 4;
          if (S.lamd > 1e12) S.stop = 1;
        }
        __syncthreads();
        if (S.stop) break;
      }
      __syncthreads();
      if (S.stop) break;
      if (!accepted)  // (40 rejected trials: the weights of the kept point again)
        cost = lm_eval(S, S.c, Eb, El, npix, cs, ce, wm, true, px);
      PP_T(11);  // (debug) accept
    }
    __syncthreads();
    if (pfit && tid < m) {
      // node values p = C c (C follows C^-1 in the Cinv buffer)
      const double *Cm = Cinv + m * m;
      double s = 0;
      for (int jj = 0; jj < m; jj++) s = fma(Cm[tid * m + jj], S.c[jj], s);
      pfit[(int64_t)b * nnode_s + tid] = s;
    }
  }

  PP_T(4);  // LM
  // ---- normalise (make_ccf.py:380-392) ----------------------------------------
  for (int k = tid; k < npix; k += PP_NT) {
    double cont = 1.0;
    if (continuum) {
      const int l = El[k];
      double s = Eb[3 * k] * S.c[l];
      s = fma(Eb[3 * k + 1], S.c[l + 1], s);
      s = fma(Eb[3 * k + 2], S.c[l + 2], s);
      cont = exp(fmin(fmax(s, -100.0), 100.0));
    }
    if (medv > 0)
      cont = fmax(1e-2 * medv, cont);
    else
      cont = fmax(cont, 1.0);
    if (cont_out) cont_out[(int64_t)b * npix_s + k] = cont;
    const double e = ce[k];
    double iv = 1.0 / (e * e);
    double c = sp0[k] / cont;
    iv = cont * cont * iv;
    if (msk[k]) {
      iv = 0;
      c = 0;
    }
    cs[k] = c;   // normalised spectrum  (every thread rewrites only the
    ce[k] = iv;  // its inverse variance   pixels it has just read)
  }
  __syncthreads();
  PP_T(5);  // normalise
  // ---- rebin to the FFT grid (make_ccf.py:394-409) -----------------------------
  double ss = 0;
  for (int n = tid; n < nfft; n += PP_NT) {
    const int xi = xind[n];
    double r1 = 0, r2 = 0;
    if (xi >= 0) {
      const double rwt = rw[n], lwt = 1 - rwt;
      r1 = lwt * cs[xi] + rwt * cs[xi + 1];
      const double a = ce[xi], c = ce[xi + 1];
      r2 = a * c / (lwt * lwt * c + rwt * rwt * a + ((a * c) == 0 ? 1.0 : 0.0));
    }
    proc_spec[(int64_t)b * nfft + n] = r1;
    proc_ivar[(int64_t)b * nfft + n] = r2;
    ss += r1 * r1 * r2;
  }
  ss = block_sum<PP_NW>(ss, S.red);
  if (tid == 0) {
    sse[b] = ss;
    if (S.flag && status) atomicOr(&status[b], S.flag);
  }
  PP_T(6);  // rebin
}

extern "C" int rvs_ccf_preprocess_g(const double *lam, const double *spec,
                                  const double *espec, const uint8_t *badmask,
                                  int npix, int B, int continuum,
                                  const double *Eb, const int32_t *El,
                                  const double *Cinv, const int32_t *istart,
                                  int nnode, const int32_t *bin_start,
                                  const int32_t *xind, const double *rw, int nfft,
                                  double maxerr, double *proc_spec,
                                  double *proc_ivar, double *sse, double *cont,
                                  double *pfit, int32_t *status,
                                  const int32_t *grid_id,
                                  const int32_t *npix_g, const int32_t *nnode_g,
                                  void *stream) {
  if (npix < 12 || B < 1 || nfft < 2) return RVS_E_ARG;
  if (grid_id && (!npix_g || (continuum && !nnode_g))) return RVS_E_ARG;
  if (continuum && (nnode < 3 || nnode > CCF_MAXNODE)) return RVS_E_ARG;
  const size_t shm = sizeof(double) * 3 * (size_t)npix + ((npix + 15) / 16) * 16;
  if (shm + sizeof(LMShared) + sizeof(SelShared) > 159 * 1024) return RVS_E_ARG;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)ccf_preprocess_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(159 * 1024 - sizeof(LMShared) - sizeof(SelShared)));
    (void)hipGetLastError();
    attr_set = true;
  }
  hipLaunchKernelGGL(ccf_preprocess_kernel, dim3(B), dim3(PP_NT), shm,
                     rvs_stream(stream), lam, spec, espec, badmask, npix,
                     continuum, Eb, El, Cinv, istart, nnode, bin_start, xind, rw,
                     nfft, maxerr, proc_spec, proc_ivar, sse, cont, pfit, status, grid_id,
                     npix_g, nnode_g);
  RVS_LAUNCH_CHECK();
  return 0;
}

extern "C" int rvs_ccf_preprocess(const double *lam, const double *spec,
                                  const double *espec, const uint8_t *badmask,
                                  int npix, int B, int continuum,
                                  const double *Eb, const int32_t *El,
                                  const double *Cinv, const int32_t *istart,
                                  int nnode, const int32_t *bin_start,
                                  const int32_t *xind, const double *rw, int nfft,
                                  double maxerr, double *proc_spec,
                                  double *proc_ivar, double *sse, double *cont,
                                  double *pfit, int32_t *status, void *stream) {
  return rvs_ccf_preprocess_g(lam, spec, espec, badmask, npix, B, continuum, Eb, El,
                              Cinv, istart, nnode, bin_start, xind, rw, nfft, maxerr,
                              proc_spec, proc_ivar, sse, cont, pfit, status, nullptr,
                              nullptr, nullptr, stream);
}

// ---------------------------------------------------------------------------
// argmin over (template, velocity), parabola refinement (fitter_ccf.py:218-236)
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool nan_less(double a, long long ka, double b,
                                         long long kb) {
  // "a before b" for numpy argmin: NaN is the minimum, first occurrence wins
  const bool an = !(a == a), bn = !(b == b);
  if (an != bn) return an;
  if (an) return ka < kb;
  return (a < b) || (a == b && ka < kb);
}

__global__ void __launch_bounds__(256)
    ccf_select_kernel(const double *__restrict__ chisq,
                      const double *__restrict__ sse, int narm, int T,
                      const double *__restrict__ vgrid, int nvel,
                      double *__restrict__ res, double *__restrict__ best_ccf,
                      int32_t *__restrict__ status) {
  __shared__ double sv[4];
  __shared__ long long sk[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  double tot = 0;
  for (int a = 0; a < narm; a++) tot += sse[(int64_t)a * gridDim.x + b];
  const double *c = chisq + (int64_t)b * T * nvel;
  // best_id = argmin_t min_v; min over a row containing NaN is NaN.  The
  // lexicographic (value, t*nvel+v) minimum gives the first template whose row
  // minimum is smallest AND, inside it, the first velocity of that minimum.
  // A NaN anywhere in a row makes that row's min NaN: handled by NaN-first order
  // with key = t*nvel (row start) so the earliest NaN row wins as in numpy.
  double bv = __builtin_inf();
  long long bk = (1ll << 62);
  for (int e = tid; e < T * nvel; e += 256) {
    const double x = c[e] + tot;
    if (nan_less(x, e, bv, bk)) {
      bv = x;
      bk = e;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(bv, o, 64);
    const long long ok = __shfl_xor(bk, o, 64);
    if (nan_less(ov, ok, bv, bk)) {
      bv = ov;
      bk = ok;
    }
  }
  if ((tid & 63) == 0) {
    sv[tid >> 6] = bv;
    sk[tid >> 6] = bk;
  }
  __syncthreads();
  bv = sv[0];
  bk = sk[0];
  for (int w = 1; w < 4; w++)
    if (nan_less(sv[w], sk[w], bv, bk)) {
      bv = sv[w];
      bk = sk[w];
    }
  const int best_id = (int)(bk / nvel);
  const double *row = c + (int64_t)best_id * nvel;
  for (int v = tid; v < nvel; v += 256)
    best_ccf[(int64_t)b * nvel + v] = row[v] + tot;
  if (tid == 0) {
    // first minimum inside the winning row (np.argmin(best_ccf))
    int bp = 0;
    double mv = row[0] + tot;
    for (int v = 1; v < nvel; v++) {
      const double x = row[v] + tot;
      if (nan_less(x, v, mv, bp)) {
        mv = x;
        bp = v;
      }
    }
    double vel = vgrid[bp];
    if (bp != 0 && bp != nvel - 1) {
      const double xa = vgrid[bp - 1], xb = vgrid[bp], xc = vgrid[bp + 1];
      const double ya = row[bp - 1] + tot, yb = row[bp] + tot, yc = row[bp + 1] + tot;
      const double d1 = (yb - ya) / (xb - xa), d2 = (yc - yb) / (xc - xb);
      const double a2 = (d2 - d1) / (xc - xa);
      if (a2 > 0) vel = xb - (d1 + a2 * (xb - xa)) / (2 * a2);
    }
    int st = 0;
    if (!(fabs(mv) <= 1.79e308)) st |= RVS_ST_CCF_FAILED;
    double *r = res + (int64_t)b * 4;
    r[0] = best_id;
    r[1] = vel;
    r[2] = bp;
    r[3] = mv;
    if (st && status) atomicOr(&status[b], st);
  }
}

extern "C" int rvs_ccf_select(const double *chisq, const double *sse,
                              int narm_sse, int B, int T, const double *vgrid,
                              int nvel, double *res, double *best_ccf,
                              int32_t *status, void *stream) {
  if (B < 1 || T < 1 || nvel < 1) return RVS_E_ARG;
  hipLaunchKernelGGL(ccf_select_kernel, dim3(B), dim3(256), 0,
                     rvs_stream(stream), chisq, sse, narm_sse, T, vgrid, nvel,
                     res, best_ccf, status);
  RVS_LAUNCH_CHECK();
  return 0;
}
