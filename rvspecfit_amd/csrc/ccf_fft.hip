// ccf_fft.hip -- FFT cross-correlation for gfx950.
//
// Reference: py/rvspecfit/fitter_ccf.py:126-161 (rfft of spec*ivar and ivar,
// lag tables) and :189-216 (irfft(F_t S*), irfft(F2_t V*), chi^2 = -2 c0 + c1 or
// -c0^2/c1, linear interpolation to the common velocity grid, sum over arms).
//
//  ccf_rfft_kernel   conj(rfft(spec*ivar)), conj(rfft(ivar)) per spectrum.
//  ccf_xcorr_kernel  one 512-thread block per (spectrum, template): the
//      template's two complex128 spectra and the spectrum's S*, V* are streamed
//      from HBM/L2 (coalesced 16 B per lane; bin pairs k, N/2-k handled together
//      so every element is read once), the Hermitian pre-processing of the
//      half-size inverse transform is done on the fly, then one N/2-point complex
//      FFT in LDS: radix-8 decimation-in-frequency passes, one butterfly per
//      thread per pass, un-padded LDS image followed by the passes' twiddles.  In continuum mode the reference's two
//      inverse transforms collapse into ONE by linearity:
//      -2 c0 + c1 = irfft(-2 F S* + F2 V*).  Only the ~100 lags inside +-max_vel
//      are gathered (digit-reversed positions come from the host through
//      rvs_ccf_fft_pos), interpolated (scipy interp1d formula) and accumulated:
//      out = beta*out + value, so arms add up without atomics.
#include "common.h"

#define XC_NT 512
// LDS image of a transform: n2 complex points, un-padded (round 1 padded the index
// 9/8 against bank conflicts of the stride-8^k passes; measured in round 2 the
// padding -- 1/8, 1/16 or none -- changes nothing (51.9 vs 52.2 ms per step), and its
// 8 KB now hold the butterfly twiddles), followed by
// XC_NTW(n2) = n2/8 twiddles  T1[i] = exp(+2 pi i (2 i) / nfft):
// every radix-8 pass reads its first twiddle w1 = T1[r << (tws - 1)] from LDS.  As
// a global load after each barrier it cost ~8 % of the kernel (the 64 KB table
// does not survive in L1 beside the operand stream).
#define XC_NTW(n2) ((n2) >> 3)

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cadd(double2 a, double2 b) {
  return make_double2(a.x + b.x, a.y + b.y);
}
__device__ __forceinline__ double2 csub(double2 a, double2 b) {
  return make_double2(a.x - b.x, a.y - b.y);
}
// multiply by +i (SIGN>0) or -i (SIGN<0)
template <int SIGN>
__device__ __forceinline__ double2 cmuli(double2 a) {
  return (SIGN > 0) ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x);
}

// radix plan of an n2 = 2^log2n point transform: 8,8,...,(4|2)
__host__ __device__ inline int xc_plan(int log2n, int *rad) {
  int n = 0, l = log2n;
  while (l >= 3) {
    rad[n++] = 8;
    l -= 3;
  }
  if (l == 2) rad[n++] = 4;
  if (l == 1) rad[n++] = 2;
  return n;
}

// position (un-padded) of output frequency f after the in-place DIF passes
extern "C" int rvs_ccf_fft_pos(int nfft, int f) {
  int log2n = 0;
  while ((2 << log2n) < nfft) log2n++;
  int rad[8];
  const int np = xc_plan(log2n, rad);
  int n2 = nfft >> 1, pos = 0, m = n2;
  for (int p = 0; p < np; p++) {
    const int q = f % rad[p];
    f /= rad[p];
    m /= rad[p];
    pos += q * m;
  }
  return pos;
}

// y_q = sum_j a_j e^{SIGN 2 pi i j q / R}, natural order in and out
template <int SIGN>
__device__ __forceinline__ void dft4(double2 &a0, double2 &a1, double2 &a2,
                                     double2 &a3) {
  const double2 t0 = cadd(a0, a2), t1 = csub(a0, a2);
  const double2 t2 = cadd(a1, a3), t3 = cmuli<SIGN>(csub(a1, a3));
  a0 = cadd(t0, t2);
  a2 = csub(t0, t2);
  a1 = cadd(t1, t3);
  a3 = csub(t1, t3);
}

template <int SIGN>
__device__ __forceinline__ void dft8(double2 *a) {
  // even / odd split, then twiddles W8^k = e^{SIGN i pi k/4}
  double2 e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6];
  double2 o0 = a[1], o1 = a[3], o2 = a[5], o3 = a[7];
  dft4<SIGN>(e0, e1, e2, e3);
  dft4<SIGN>(o0, o1, o2, o3);
  const double h = 0.70710678118654752440;
  // W8^1 = (1 + SIGN i) h ; W8^2 = SIGN i ; W8^3 = (-1 + SIGN i) h
  const double2 w1o = (SIGN > 0) ? make_double2(h * (o1.x - o1.y), h * (o1.x + o1.y))
                                 : make_double2(h * (o1.x + o1.y), h * (o1.y - o1.x));
  const double2 w2o = cmuli<SIGN>(o2);
  const double2 w3o = (SIGN > 0) ? make_double2(-h * (o3.x + o3.y), h * (o3.x - o3.y))
                                 : make_double2(h * (o3.y - o3.x), -h * (o3.x + o3.y));
  a[0] = cadd(e0, o0);
  a[4] = csub(e0, o0);
  a[1] = cadd(e1, w1o);
  a[5] = csub(e1, w1o);
  a[2] = cadd(e2, w2o);
  a[6] = csub(e2, w2o);
  a[3] = cadd(e3, w3o);
  a[7] = csub(e3, w3o);
}

// tw[j] = exp(+2 pi i j/nfft), j < nfft/2; returns exp(SIGN 2 pi i e / nfft)
template <int SIGN>
__device__ __forceinline__ double2 twid(const double2 *__restrict__ tw, int e,
                                        int nfft) {
  const int half = nfft >> 1;
  double2 w = (e < half) ? tw[e] : tw[e - half];
  if (e >= half) {
    w.x = -w.x;
    w.y = -w.y;
  }
  if (SIGN < 0) w.y = -w.y;
  return w;
}

#ifdef RVS_XC_TIMING
// debug build only (tools/perf/xc_phases.sh): clock budget of ccf_xcorr_kernel
__device__ unsigned long long xc_dbg[16];
#define XC_T(i)                                        \
  do {                                                 \
    __syncthreads();                                   \
    if (threadIdx.x == 0) {                            \
      const unsigned long long t_ = wall_clock64();    \
      atomicAdd(&xc_dbg[i], t_ - t_prev);              \
      t_prev = t_;                                     \
    }                                                  \
  } while (0)
extern "C" int rvs_dbg_read_xc(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(xc_dbg), sizeof(xc_dbg)) ==
                 hipSuccess
             ? 0
             : -1;
}
#else
#define XC_T(i)
#endif

// T1 of the image a[] (see XC_NTW); visible after the next barrier.
// LDS-DMA (global_load_lds_dwordx4: no register, nothing to wait for until that
// barrier) when a whole number of waves covers the table: as `load; wait;
// ds_write` at the top of the kernel the fill cost every wave an L2 round trip
// before its first operand load.  Wave w writes T1[64 w .. 64 w + 63]: the LDS
// destination of a DMA is wave-uniform base + lane x 16 bytes.
template <int NT>
__device__ __forceinline__ void xc_fill_twiddles(double2 *a, int n2,
                                                 const double2 *__restrict__ tw) {
  const int ntw = XC_NTW(n2);
  if ((ntw & 63) == 0) {
    const int lane = threadIdx.x & 63;
    for (int i0 = (threadIdx.x >> 6) * 64; i0 < ntw; i0 += NT) {
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)(tw + 2 * (i0 + lane)),
          (__attribute__((address_space(3))) void *)(a + n2 + i0), 16, 0, 0);
    }
    return;
  }
  for (int i = threadIdx.x; i < ntw; i += NT) a[n2 + i] = tw[2 * i];
}

// in-place DIF transform of the LDS image a[] (n2 points)
// prune (nullable): output masks for the LAST TWO passes when both are radix 8
// (n2 a power of 8).  Only ~100 of the n2 outputs of the inverse transform are
// ever read (the lags inside +-max_vel), so
//   pass np-2: prune[blk]          bit q set -> output q of every butterfly of
//              64-block blk is needed (the others are neither twiddled nor
//              stored: 1-2 writes and twiddles instead of 8 and 7),
//   pass np-1: prune[n2/64 + grp]  bit q set -> output q of 8-group grp is needed
//              (groups with an empty mask are skipped: ~64 of 512 remain).
// a thread's share of the output masks, fetched by the caller ahead of the operand
// stream (a load right before the passes is an exposed L2 round trip per block):
// g = mask of 8-group threadIdx.x, b0 / b7 = masks of the first and last 8-group
// of 64-block threadIdx.x >> 3 (every n2 that can be pruned -- 64, 512, 4096 --
// has at most NT such groups and NT / 8 such blocks)
struct XcMasks {
  unsigned g, b0, b7;
};
template <int NT>
__device__ __forceinline__ XcMasks xc_load_masks(const uint8_t *__restrict__ prune,
                                                 int n2) {
  XcMasks m = {0u, 0u, 0u};
  if (prune && (n2 >> 3) <= NT) {
    const int t = threadIdx.x, B = t >> 3;
    if (t < (n2 >> 3)) m.g = prune[(n2 >> 6) + t];
    if (B < (n2 >> 6)) {
      m.b0 = prune[(n2 >> 6) + 8 * B];
      m.b7 = prune[(n2 >> 6) + 8 * B + 7];
    }
  }
  return m;
}

template <int SIGN, int NT>
__device__ void fft_lds(double2 *a, int log2n, const double2 *__restrict__ tw,
                        const uint8_t *__restrict__ prune = nullptr,
                        const XcMasks *pm = nullptr) {
  const int n2 = 1 << log2n, nfft = n2 << 1;
  const double2 *T1 = a + n2;  // filled by xc_fill_twiddles before the first pass
  auto tw1 = [&](int r, int tws) -> double2 {
    double2 w = T1[r << (tws - 1)];
    if (SIGN < 0) w.y = -w.y;
    return w;
  };
  int rad[8];
  const int np = xc_plan(log2n, rad);
  const bool pr = prune && np >= 2 && rad[np - 1] == 8 && rad[np - 2] == 8;
  // The lags inside +-max_vel are a few dozen outputs either side of zero: of every
  // 64-block that enters the last two passes only frequency 0 (position 0) or 63
  // (position 63) is read back -- X[0] = sum_n y[n], X[63] = sum_n y[n] W64^(63 n).
  // When the masks say so (checked here, one 8-group per thread), the two pruned
  // passes collapse into one without a barrier in between: eight lanes per block,
  // lane s takes y[s + 8 j], j < 8, through the radix-8 butterfly (outputs 0 and
  // 7 are what the masked pass np-2 kept), applies W64^(63 s) from the twiddle
  // table and the eight lanes are summed with DPP -- instead of a store, a
  // barrier, and a last pass in which one wave of eight worked.
  bool fold = false;
  const bool can_fold = pr && pm && (n2 >> 3) <= NT;
  int lgM = log2n;
  for (int p = 0; p < np; p++) {
    if (can_fold && p == 0) {
      // (the first pass's barrier doubles as the block-wide vote)
      const int pos8 = threadIdx.x & 7;
      const unsigned m = pm->g;
      const bool bad =
          m != 0 && !((pos8 == 0 && m == 0x01u) || (pos8 == 7 && m == 0x80u));
      fold = !__syncthreads_or(bad);
    }
    if (fold && p == np - 2) {
      __syncthreads();
      const int s = threadIdx.x & 7;
      for (int B = threadIdx.x >> 3; B < (n2 >> 6); B += NT >> 3) {
        const unsigned m0 = pm->b0 & 1u, m63 = pm->b7 >> 7;
        if (!(m0 | m63)) continue;   // (uniform over the eight lanes of a block)
        double2 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = a[B * 64 + s + 8 * j];
        dft8<SIGN>(v);
        // W64^(63 s) = conj(W64^s), W64^s = exp(SIGN 2 pi i s (nfft/64) / nfft)
        double2 w = T1[s << (log2n - 6)];
        if (SIGN > 0) w.y = -w.y;
        double2 x63 = cmul(v[7], w), x0 = v[0];
        // sum over the eight lanes: xor 1, xor 2 (quad_perm), mirror of 8
        auto sum8 = [](double x) {
          x += dpp_get<0xb1, 0xf, 0xf>(x);    // quad_perm [1,0,3,2]
          x += dpp_get<0x4e, 0xf, 0xf>(x);    // quad_perm [2,3,0,1]
          x += dpp_get<0x141, 0xf, 0xf>(x);   // row_half_mirror
          return x;
        };
        if (m0) {
          x0.x = sum8(x0.x);
          x0.y = sum8(x0.y);
          if (s == 0) a[B * 64] = x0;
        }
        if (m63) {
          x63.x = sum8(x63.x);
          x63.y = sum8(x63.y);
          if (s == 0) a[B * 64 + 63] = x63;
        }
      }
      break;
    }
    // every size is a power of two: shifts, not the ~30-instruction integer
    // divisions a runtime divisor costs each thread in each pass
    const int R = rad[p], lgR = (R == 8) ? 3 : (R == 4 ? 2 : 1);
    const int lgMp = lgM - lgR, Mp = 1 << lgMp;
    const int tws = log2n + 1 - lgM;  // twiddle stride nfft / M = 1 << tws
    if (!(can_fold && p == 0)) __syncthreads();
    for (int u = threadIdx.x; u < (n2 >> lgR); u += NT) {
      const int blk = u >> lgMp, r = u & (Mp - 1);
      const int base = (blk << lgM) + r;
      if (pr && p == np - 2) {  // M = 64, Mp = 8: selected outputs only
        const unsigned mask = prune[blk];
        if (mask == 0) continue;
        double2 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = a[base + Mp * j];
        dft8<SIGN>(v);
        // the same product tree for the twiddle powers as the full pass, so
        // the stored values are bit-identical to it
        double2 wp[8];
        wp[0] = make_double2(1.0, 0.0);
        wp[1] = tw1(r, tws);
        wp[2] = cmul(wp[1], wp[1]);
        wp[3] = cmul(wp[2], wp[1]);
        wp[4] = cmul(wp[2], wp[2]);
        wp[7] = cmul(wp[4], wp[3]);
        wp[5] = cmul(wp[4], wp[1]);
        wp[6] = cmul(wp[4], wp[2]);
#pragma unroll
        for (int q = 0; q < 8; q++)
          if (mask & (1u << q))
            a[base + Mp * q] = (q == 0) ? v[0] : cmul(v[q], wp[q]);
        continue;
      }
      if (pr && p == np - 1) {  // M = 8, Mp = 1: no twiddles
        const unsigned mask = prune[(n2 >> 6) + u];
        if (mask == 0) continue;
        double2 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = a[base + j];
        dft8<SIGN>(v);
#pragma unroll
        for (int q = 0; q < 8; q++)
          if (mask & (1u << q)) a[base + q] = v[q];
        continue;
      }
      if (R == 8) {
        double2 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = a[base + Mp * j];
        dft8<SIGN>(v);
        if (Mp > 1) {
          // powers of w1 generated and consumed one at a time (short live
          // ranges: 3 complex instead of 7); every power is a product of at most
          // two squarings/multiplications of table values
          const double2 w1 = tw1(r, tws);
          v[1] = cmul(v[1], w1);
          const double2 w2 = cmul(w1, w1);
          v[2] = cmul(v[2], w2);
          double2 wc = cmul(w2, w1);  // w3
          v[3] = cmul(v[3], wc);
          const double2 w4 = cmul(w2, w2);
          v[4] = cmul(v[4], w4);
          v[7] = cmul(v[7], cmul(w4, wc));  // w7 = w4 w3
          wc = cmul(w4, w1);                // w5
          v[5] = cmul(v[5], wc);
          v[6] = cmul(v[6], cmul(w4, w2));  // w6
        }
#pragma unroll
        for (int q = 0; q < 8; q++) a[base + Mp * q] = v[q];
      } else if (R == 4) {
        double2 v0 = a[base], v1 = a[base + Mp],
                v2 = a[base + 2 * Mp], v3 = a[base + 3 * Mp];
        dft4<SIGN>(v0, v1, v2, v3);
        if (Mp > 1) {
          const double2 w1 = twid<SIGN>(tw, r << tws, nfft);
          const double2 w2 = cmul(w1, w1), w3 = cmul(w2, w1);
          v1 = cmul(v1, w1);
          v2 = cmul(v2, w2);
          v3 = cmul(v3, w3);
        }
        a[base] = v0;
        a[base + Mp] = v1;
        a[base + 2 * Mp] = v2;
        a[base + 3 * Mp] = v3;
      } else {
        const double2 v0 = a[base], v1 = a[base + Mp];
        double2 d = csub(v0, v1);
        if (Mp > 1) d = cmul(d, twid<SIGN>(tw, r << tws, nfft));
        a[base] = cadd(v0, v1);
        a[base + Mp] = d;
      }
    }
    lgM = lgMp;
  }
  __syncthreads();
}

// (device form of rvs_ccf_fft_pos; the radices are powers of two: masks and
// shifts -- with `%` and `/` by the run-time radix this was ~60 integer divisions
// per thread of ccf_rfft_kernel, most of its time)
__device__ __forceinline__ int fft_pos(int f, int log2n) {
  int pos = 0, lgm = log2n;
  while (lgm > 0) {
    const int lgr = lgm >= 3 ? 3 : lgm;  // plan 8,8,...,(4|2)
    lgm -= lgr;
    pos += (f & ((1 << lgr) - 1)) << lgm;
    f >>= lgr;
  }
  return pos;
}

// conj(rfft(x)) for x = proc_spec*proc_ivar (which 0) and proc_ivar (which 1)
__global__ void __launch_bounds__(XC_NT)
    ccf_rfft_kernel(const double *__restrict__ proc_spec,
                    const double *__restrict__ proc_ivar, int nfft, int log2n,
                    const double2 *__restrict__ tw, double2 *__restrict__ work) {
  extern __shared__ double2 fa[];
  const int b = blockIdx.x, which = blockIdx.y, tid = threadIdx.x;
  const int n2 = nfft >> 1;
  const double *ps = proc_spec + (int64_t)b * nfft;
  const double *pi = proc_ivar + (int64_t)b * nfft;
  xc_fill_twiddles<XC_NT>(fa, n2, tw);
  for (int n = tid; n < n2; n += XC_NT) {
    double x0, x1;
    if (which == 0) {
      x0 = ps[2 * n] * pi[2 * n];
      x1 = ps[2 * n + 1] * pi[2 * n + 1];
    } else {
      x0 = pi[2 * n];
      x1 = pi[2 * n + 1];
    }
    fa[n] = make_double2(x0, x1);
  }
  fft_lds<-1, XC_NT>(fa, log2n, tw);
  double2 *out = work + ((int64_t)b * 2 + which) * (n2 + 1);
  for (int k = tid; k <= n2; k += XC_NT) {
    double2 X;
    if (k == 0 || k == n2) {
      const double2 z0 = fa[0];
      X = make_double2(k == 0 ? z0.x + z0.y : z0.x - z0.y, 0.0);
    } else {
      const double2 zk = fa[fft_pos(k, log2n)];
      const double2 zm = fa[fft_pos(n2 - k, log2n)];
      const double2 e = make_double2(zk.x + zm.x, zk.y - zm.y);   // zk + conj(zm)
      const double2 d = make_double2(zk.x - zm.x, zk.y + zm.y);   // zk - conj(zm)
      const double2 q = cmul(twid<-1>(tw, k, nfft), d);
      X = make_double2(0.5 * (e.x + q.y), 0.5 * (e.y - q.x));     // e/2 - (i/2) q
    }
    out[k] = make_double2(X.x, -X.y);  // conjugate
  }
}

#define XB_NT 512  // threads per block

// (spectrum, template) of a block when the template set is large.  Blocks are
// dealt round-robin over the 8 XCDs, each with its own 4 MB L2.  In the plain
// order (t fastest) an XCD meets every template again and again, which is fine
// while the set (T x 128 KB per arm) is a few MB: T = 76, 9.7 MB -> 45.6 ms per
// step against 48.0 ms with the map below.  For a large set (T = 534: 68 MB) XCD
// x owns the templates t = x (mod 8) and walks them in groups of G -- for every
// group all B spectra, for every spectrum the group's templates -- so that a
// group's template spectra (G x 128 KB, ~2 MB) stay resident in that L2 while
// the spectra stream through it once per group: 323 -> 312 ms per step.
// Launch xc_nblocks() blocks; false = padding.
__device__ __forceinline__ bool xc_job(unsigned bid, int T, int B, int G, int &b,
                                       int &t) {
  const int xcd = bid & 7;
  const int slot = bid >> 3;
  const int tt = slot % G;
  const int q = slot / G;
  b = q % B;
  const int g = q / B;
  t = xcd + 8 * (g * G + tt);
  return t < T;
}
__host__ inline int64_t xc_nblocks(int T, int B, int G) {
  const int ntx = (T + 7) >> 3;
  const int ngrp = (ntx + G - 1) / G;
  return 8ll * ngrp * G * B;
}
// group size: about 2 MB of template spectra per XCD; 0 = plain order (the whole
// set is below 16 MB)
__host__ inline int xc_group(int T, int nfft) {
  const int64_t per_t = 2ll * ((nfft >> 1) + 1) * 16;
  if (per_t * T <= (16ll << 20)) return 0;
#ifndef XC_GROUP_BYTES
#define XC_GROUP_BYTES (2ll << 20)
#endif
  int G = (int)(XC_GROUP_BYTES / per_t);
  const int ntx = (T + 7) >> 3;
  if (G < 1) G = 1;
  if (G > ntx) G = ntx;
  return G;
}

// One block per (spectrum b, template t); two blocks are resident per CU (LDS
// 74 kB each, < 128 VGPRs) so that the load phase of one overlaps the LDS / FFT
// phase of the other.  Measured alternatives: 256-thread
// blocks 17.8 ms, 512-thread blocks 14.4 ms per 2000 DESI spectra; keeping the
// spectrum's S*, V* in registers across a chunk of templates halves the L2->CU
// traffic but needs > 180 VGPRs (one block per CU or spills): 25-38 ms.
__global__ void __launch_bounds__(XB_NT)
    __attribute__((amdgpu_waves_per_eu(4, 4)))  // two 8-wave blocks per CU
    ccf_xcorr_kernel(const double2 *__restrict__ work, int nfft, int log2n,
                     const double2 *__restrict__ tfft,
                     const double2 *__restrict__ tfft2, int T,
                     const double2 *__restrict__ tw, int continuum,
                     const int32_t *__restrict__ lag_pos,
                     const double *__restrict__ lag_vel, int nlag,
                     const int32_t *__restrict__ ilo,
                     const double *__restrict__ vgrid, int nvel, double beta,
                     const uint8_t *__restrict__ prune,
                     double *__restrict__ chisq, int B, int G) {
  extern __shared__ double2 fa[];
  const int n2 = nfft >> 1, npair = n2 >> 1;
  double *c0 = reinterpret_cast<double *>(fa + n2 + XC_NTW(n2));  // [nlag]
  double *c1 = c0 + nlag;                                        // [nlag]
  int t = blockIdx.x, b = blockIdx.y;
  if (G > 0 && !xc_job(blockIdx.x, T, B, G, b, t)) return;
  const int tid = threadIdx.x;
  const double2 *Sc = work + ((int64_t)b * 2) * (n2 + 1);
  const double2 *Vc = Sc + (n2 + 1);
  const double2 *F = tfft + (int64_t)t * (n2 + 1);
  const double2 *F2 = tfft2 + (int64_t)t * (n2 + 1);
  const double inv_n = 1.0 / nfft;
  const int npass = continuum ? 1 : 2;
  xc_fill_twiddles<XB_NT>(fa, n2, tw);
  XcMasks pmask = xc_load_masks<XB_NT>(prune, n2);
#ifdef RVS_XC_TIMING
  unsigned long long t_prev = wall_clock64();
#endif
  // The read-back at the end of the block (lag gather, then interpolation onto
  // the velocity grid, then `out = beta out + value`) was three dependent global
  // round trips with most of the block idle.  Everything in it that does not
  // depend on the transform is fetched now, into registers, while the operand
  // stream is in flight (the common case nlag, nvel <= 512: one item per thread).
  double *out = chisq + ((int64_t)b * T + t) * nvel;
  const bool pre = (nlag <= XB_NT) && (nvel <= XB_NT);
  int pre_pos = 0, pre_lo = 0;
  double pre_x0 = 0, pre_x1 = 1, pre_xg = 0, pre_old = 0;
  // (clamped indices, not branches: a conditional load into a register that has
  // another definition is waited for at the join.  The second level of the
  // interpolation table, lag_vel[pre_lo], follows after the operand loop: asked
  // for here it made every wave wait for ilo before its first operand load)
  if (pre) {
    const int tv = min(tid, nvel - 1);
    pre_pos = lag_pos[min(tid, nlag - 1)];
    pre_lo = ilo[tv];
    pre_xg = vgrid[tv];
    pre_old = out[tv];   // (used with beta != 0 only)
  }
  for (int pass = 0; pass < npass; pass++) {
    if (continuum) {
      // Operand streaming is latency-bound (the operands come out of L2 and a
      // wave had 9 loads in flight): bins k and k + XB_NT are fetched together,
      // 18 x 16 B per lane, through buffer loads -- one 32-bit offset register
      // serves the four arrays (F, S*, F2, V*) that are read at the same bin, so
      // the batch fits the register budget of two blocks per CU.  Bins
      // 1..npair take exactly npair / (2 XB_NT) trips (nfft = 8192: two); the
      // DC/Nyquist pair k = 0 is wave-uniform and goes through the scalar cache.
      typedef int v4i_t __attribute__((ext_vector_type(4)));
      const int nbytes = (n2 + 1) * 16;
      const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc(
          (void *)F, 0, nbytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rF2 = __builtin_amdgcn_make_buffer_rsrc(
          (void *)F2, 0, nbytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(
          (void *)Sc, 0, nbytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(
          (void *)Vc, 0, nbytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(
          (void *)tw, 0, (npair + 1) * 16, 0x00020000);
      auto ld = [](const __amdgpu_buffer_rsrc_t &r, int off) {
        const v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
        return make_double2(__hiloint2double(v.y, v.x),
                            __hiloint2double(v.w, v.z));
      };
      if (tid == 0) {
        const double2 p1 = cmul(F[0], Sc[0]), p2 = cmul(F2[0], Vc[0]);
        const double2 q1 = cmul(F[n2], Sc[n2]), q2 = cmul(F2[n2], Vc[n2]);
        const double xk = p2.x - 2 * p1.x, xm = q2.x - 2 * q1.x;
        // numpy irfft ignores the imaginary parts of the DC and Nyquist bins
        fa[0] = make_double2(xk + xm, xk - xm);
      }
      for (int k0 = 1 + tid; k0 <= npair; k0 += 2 * XB_NT) {
        double2 op[2][9];
        bool live[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int kk = k0 + u * XB_NT;
          live[u] = kk <= npair;
          const int k = live[u] ? kk : k0;
          const int ok = k * 16, om = (n2 - k) * 16;
          op[u][0] = ld(rF, ok), op[u][1] = ld(rS, ok);
          op[u][2] = ld(rF2, ok), op[u][3] = ld(rV, ok);
          op[u][4] = ld(rF, om), op[u][5] = ld(rS, om);
          op[u][6] = ld(rF2, om), op[u][7] = ld(rV, om);
          op[u][8] = ld(rT, ok);
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int k = k0 + u * XB_NT, m = n2 - k;
          const double2 p1 = cmul(op[u][0], op[u][1]),
                        p2 = cmul(op[u][2], op[u][3]);
          const double2 Xk = make_double2(p2.x - 2 * p1.x, p2.y - 2 * p1.y);
          const double2 q1 = cmul(op[u][4], op[u][5]),
                        q2 = cmul(op[u][6], op[u][7]);
          const double2 Xm = make_double2(q2.x - 2 * q1.x, q2.y - 2 * q1.y);
          const double2 e = make_double2(Xk.x + Xm.x, Xk.y - Xm.y);
          const double2 d = make_double2(Xk.x - Xm.x, Xk.y + Xm.y);
          const double2 q = cmul(op[u][8], d);
          if (live[u]) {
            fa[k] = make_double2(e.x - q.y, e.y + q.x);
            if (m != k) fa[m] = make_double2(e.x + q.y, -e.y + q.x);
          }
        }
      }
    } else
    for (int k = tid; k <= npair; k += XB_NT) {
      const int m = n2 - k;
      double2 Xk, Xm;
      if (continuum) {
        const double2 p1 = cmul(F[k], Sc[k]), p2 = cmul(F2[k], Vc[k]);
        Xk = make_double2(p2.x - 2 * p1.x, p2.y - 2 * p1.y);
        const double2 q1 = cmul(F[m], Sc[m]), q2 = cmul(F2[m], Vc[m]);
        Xm = make_double2(q2.x - 2 * q1.x, q2.y - 2 * q1.y);
      } else if (pass == 0) {
        Xk = cmul(F[k], Sc[k]);
        Xm = cmul(F[m], Sc[m]);
      } else {
        Xk = cmul(F2[k], Vc[k]);
        Xm = cmul(F2[m], Vc[m]);
      }
      if (k == 0) {
        // numpy irfft ignores the imaginary parts of the DC and Nyquist bins
        fa[0] = make_double2(Xk.x + Xm.x, Xk.x - Xm.x);
      } else {
        const double2 e = make_double2(Xk.x + Xm.x, Xk.y - Xm.y);  // Xk + conj Xm
        const double2 d = make_double2(Xk.x - Xm.x, Xk.y + Xm.y);  // Xk - conj Xm
        const double2 q = cmul(tw[k], d);
        // Z[k] = e + i q ; Z[m] = conj(e) + i conj(q)  (m == k: both the same)
        fa[k] = make_double2(e.x - q.y, e.y + q.x);
        if (m != k) fa[m] = make_double2(e.x + q.y, -e.y + q.x);
      }
    }
    if (pass == 0 && pre) {
      // (the fetched-ahead values are "defined" here for the compiler: left free
      // it hoists the first trivial operation on each -- a mask bit, a sign
      // extension -- to right behind the load and waits there)
      asm volatile("" : "+v"(pre_pos), "+v"(pre_lo));
      asm volatile("" : "+v"(pmask.g), "+v"(pmask.b0), "+v"(pmask.b7));
      pre_x0 = lag_vel[pre_lo];
      pre_x1 = lag_vel[pre_lo + 1];
    }
    XC_T(0);  // operand stream + products + Hermitian fold into LDS
    fft_lds<1, XB_NT>(fa, log2n, tw, prune, &pmask);
    XC_T(1);  // the four radix-8 passes
    const double *fr = reinterpret_cast<const double *>(fa);
    double *dst = (pass == 0) ? c0 : c1;
    if (pre) {
      if (tid < nlag) dst[tid] = fr[pre_pos] * inv_n;
    } else {
      for (int l = tid; l < nlag; l += XB_NT) dst[l] = fr[lag_pos[l]] * inv_n;
    }
    __syncthreads();
  }
  if (!continuum) {
    for (int l = tid; l < nlag; l += XB_NT) c0[l] = -c0[l] * c0[l] / c1[l];
    __syncthreads();
  }
  if (pre) {
    if (tid < nvel) {
      const double sl = (c0[pre_lo + 1] - c0[pre_lo]) / (pre_x1 - pre_x0);
      const double val = sl * (pre_xg - pre_x0) + c0[pre_lo];
      out[tid] = (beta != 0.0) ? beta * pre_old + val : val;
    }
    XC_T(2);  // lag gather + interpolation + store
    return;
  }
  for (int v = tid; v < nvel; v += XB_NT) {
    const int lo = ilo[v];
    const double x0 = lag_vel[lo], x1 = lag_vel[lo + 1];
    const double sl = (c0[lo + 1] - c0[lo]) / (x1 - x0);
    const double val = sl * (vgrid[v] - x0) + c0[lo];
    out[v] = (beta != 0.0) ? beta * out[v] + val : val;
  }
}


// ---------------------------------------------------------------------------
// Wave-specialised form (round 4, the review's item 5): ONE persistent 1024-thread
// block per spectrum walks the T templates.  Waves 8-15 (producers) hold the
// spectrum's S*, V* in registers for the whole walk (4 bin pairs = 16 complex128 =
// 64 VGPRs per lane), stream the template's F, F2, form -2 F S* + F2 V* with the
// Hermitian fold and write LDS image (t + 1) & 1, while waves 0-7 (consumers) run
// the radix-8 passes, the lag read-back and the interpolation of template t on
// image t & 1.  Operands per (spectrum, template): 131 KB instead of 270 KB through
// L2 -> L1.  gfx950 has one s_barrier per workgroup: both roles execute the same
// NBAR barriers per template (after pass 0, pass 1, the folded pass [or passes 2, 3],
// the read-back); a producer's four pair batches fall into the first four intervals,
// each batch requested one interval ahead (in flight across a barrier -- the barrier
// does not wait for loads).  Same formulas per bin and per butterfly as
// ccf_xcorr_kernel (outputs equal to a few ulp: the compiler fuses the products of a
// complex multiplication its own way in each kernel).  nfft = 8192, continuum mode,
// nlag and nvel <= 512, any T (rvs_ccf_xcorr decides; RVS_XC_WS=0: the per-pair
// kernel).  Measured: 45.3 -> 33.0 ms per step at T = 76, 317 -> 268 ms at T = 534;
// consumers alone 24.8 ms, producers alone 22.7 ms; any scratch in the producers
// (resident fold twiddles: 48 B) costs more than it saves (36.6 ms).
// ---------------------------------------------------------------------------
#define XW_NT 1024
#define XW_HALF 512

// radix-8 DIF pass p (M = n2 >> 3p) of the n2 = 2^LOG2N point image a[]: butterfly t
// (Tc, nullable: the pass's twiddles T1[r << (tws - 1)], r < Mp, copied side by side --
// read at their stride in T1 the lanes of a wave land on two bank groups: pass 1 of
// the 4096-point image reads 64 twiddles 128 bytes apart, a 32-way conflict per wave
// and template: profiles/r05_sq_counters.json, bank-conflict cycles 23 % of LDS-active)
template <int SIGN, int LOG2N>
__device__ __forceinline__ void xw_pass(double2 *a, const double2 *T1, int p, int t,
                                        const double2 *Tc = nullptr) {
  const int lgM = LOG2N - 3 * p, lgMp = lgM - 3, Mp = 1 << lgMp;
  const int tws = LOG2N + 1 - lgM;
  const int blk = t >> lgMp, r = t & (Mp - 1);
  const int base = (blk << lgM) + r;
  double2 v[8];
#pragma unroll
  for (int j = 0; j < 8; j++) v[j] = a[base + Mp * j];
  dft8<SIGN>(v);
  if (Mp > 1) {
    double2 w1 = Tc ? Tc[r] : T1[r << (tws - 1)];
    if (SIGN < 0) w1.y = -w1.y;
    v[1] = cmul(v[1], w1);
    const double2 w2 = cmul(w1, w1);
    v[2] = cmul(v[2], w2);
    double2 wc = cmul(w2, w1);  // w3
    v[3] = cmul(v[3], wc);
    const double2 w4 = cmul(w2, w2);
    v[4] = cmul(v[4], w4);
    v[7] = cmul(v[7], cmul(w4, wc));  // w7 = w4 w3
    wc = cmul(w4, w1);                // w5
    v[5] = cmul(v[5], wc);
    v[6] = cmul(v[6], cmul(w4, w2));  // w6
  }
#pragma unroll
  for (int q = 0; q < 8; q++) a[base + Mp * q] = v[q];
}

// LOG2N = 12 (nfft 8192: passes 8,8,8,8, the last two pruned / folded) or 11 (nfft
// 4096: passes 8,8,8,4, nothing pruned).  NBAR barriers per template: 4 (12, folded) or 5.
// RATIO: the mode without continuum normalisation, -c0^2 / c1 at the lags
// (fitter_ccf.py:204-207), is not linear in the two correlations: an iteration takes ONE
// of them -- image 2t is F_t S*, image 2t + 1 is F2_t V* (a producer forms one product
// per bin pair and requests half the operands), the read-back of image 2t parks c0 at
// the lags, the read-back of 2t + 1 forms -c0^2 / c1 there (the lane that parked c0[l]
// is the one that reads it) and only then the interpolation runs.  2 T iterations, the
// same passes, barriers and formulas per bin as ccf_xcorr_kernel's two passes.
template <int LOG2N, bool RATIO>
__global__ void __launch_bounds__(XW_NT)
    ccf_xcorr_ws_kernel(const double2 *__restrict__ work,
                        const double2 *__restrict__ tfft,
                        const double2 *__restrict__ tfft2, int T,
                        const double2 *__restrict__ tw,
                        const int32_t *__restrict__ lag_pos,
                        const double *__restrict__ lag_vel, int nlag,
                        const int32_t *__restrict__ ilo,
                        const double *__restrict__ vgrid, int nvel, double beta,
                        const uint8_t *__restrict__ prune,
                        double *__restrict__ chisq) {
  extern __shared__ double2 fa[];
  constexpr int log2n = LOG2N, n2 = 1 << LOG2N, nfft = 2 * n2, npair = n2 / 2;
  constexpr int NPP = npair / XW_HALF;   // bin pairs of a producer lane: 4 or 2
  constexpr bool P12 = (LOG2N == 12);
  static_assert(LOG2N == 12 || LOG2N == 11, "plans 8,8,8,8 and 8,8,8,4 only");
  auto img = [&](int i) -> double2 * { return fa + (i & 1) * n2; };
  const double2 *T1 = fa + 2 * n2;                         // [n2 / 8]
  double *c0 = reinterpret_cast<double *>(fa + 2 * n2 + XC_NTW(n2));  // [nlag]
  // pass 1's 64 twiddles and the folded pass's 8, side by side (xw_pass)
  double2 *T1c = fa + 2 * n2 + XC_NTW(n2) + ((nlag + 1) >> 1);   // [64 + 8]
  double *cA = reinterpret_cast<double *>(T1c + 73);   // RATIO: c0 at the lags [nlag]
  const int TI = RATIO ? 2 * T : T;   // image iterations
  const int b = blockIdx.x, tid = threadIdx.x;
  const bool producer = tid >= XW_HALF;
  const int pt = tid & (XW_HALF - 1);
  const double inv_n = 1.0 / nfft;
  if (P12 && tid < 72)   // (straight from the table: the LDS copy of T1 is in flight)
    T1c[tid] = tid < 64 ? tw[2 * (tid << 3)] : tw[2 * ((tid - 64) << (log2n - 6))];
  xc_fill_twiddles<XW_NT>(fa + n2, n2, tw);   // -> fa[2 n2 + i]
  // output masks of the last two passes (consumers), and the block-wide vote on the
  // folded form of those passes (see fft_lds)
  XcMasks pmask = {0u, 0u, 0u};
  bool bad = !P12;
  if (P12 && !producer) {
    pmask = xc_load_masks<XW_HALF>(prune, n2);
    const int pos8 = pt & 7;
    const unsigned m = pmask.g;
    bad = m != 0 && !((pos8 == 0 && m == 0x01u) || (pos8 == 7 && m == 0x80u));
  }
  const bool fold = !__syncthreads_or(bad);
  const double2 *Sc = work + ((int64_t)b * 2) * (n2 + 1);
  const double2 *Vc = Sc + (n2 + 1);
  typedef int v4i_t __attribute__((ext_vector_type(4)));
  auto ld = [](const __amdgpu_buffer_rsrc_t &r, int off) {
    const v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_double2(__hiloint2double(v.y, v.x), __hiloint2double(v.w, v.z));
  };
  const int nbytes = (n2 + 1) * 16;
  if (producer) {
    // ---- producers -------------------------------------------------------
    double2 Sk[NPP], Vk[NPP], Sm[NPP], Vm[NPP];
#pragma unroll
    for (int u = 0; u < NPP; u++) {
      const int k = 1 + pt + u * XW_HALF, m = n2 - k;
      Sk[u] = Sc[k], Vk[u] = Vc[k], Sm[u] = Sc[m], Vm[u] = Vc[m];
    }
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(
        (void *)tw, 0, (npair + 1) * 16, 0x00020000);
    double2 op[2][5];   // the batch in flight / the batch being consumed
    auto issue = [&](int t, int u, double2 *o) {
      const int tt = RATIO ? (t >> 1) : t;
      const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc(
          (void *)(((RATIO && (t & 1)) ? tfft2 : tfft) + (int64_t)tt * (n2 + 1)), 0,
          nbytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rF2 = __builtin_amdgcn_make_buffer_rsrc(
          (void *)(tfft2 + (int64_t)tt * (n2 + 1)), 0, nbytes, 0x00020000);
      // (the lane's bin offsets are re-derived at every use: kept live across the
      // template loop they are the registers that spill)
      int ptv = pt;
      asm volatile("" : "+v"(ptv));
      const int k = 1 + ptv + u * XW_HALF;
      const int ok = k * 16, om = (n2 - k) * 16;
      if (RATIO) {   // (one of the two products: F or F2 through rF)
        o[0] = ld(rF, ok);
        o[2] = ld(rF, om);
      } else {
        o[0] = ld(rF, ok), o[1] = ld(rF2, ok);
        o[2] = ld(rF, om), o[3] = ld(rF2, om);
      }
      o[4] = ld(rT, ok);
    };
    double2 dc[4];   // F[0], F2[0], F[n2], F2[n2] of the template being formed (scalar)
    auto form = [&](double2 *dst, int u, const double2 *o, int tcur) {
      int ptv = pt;
      asm volatile("" : "+v"(ptv));
      const int k = 1 + ptv + u * XW_HALF, m = n2 - k;
      double2 Xk, Xm;
      if (RATIO) {
        const bool second = (tcur & 1) != 0;
        Xk = cmul(o[0], second ? Vk[u] : Sk[u]);
        Xm = cmul(o[2], second ? Vm[u] : Sm[u]);
      } else {
        const double2 p1 = cmul(o[0], Sk[u]), p2 = cmul(o[1], Vk[u]);
        Xk = make_double2(p2.x - 2 * p1.x, p2.y - 2 * p1.y);
        const double2 q1 = cmul(o[2], Sm[u]), q2 = cmul(o[3], Vm[u]);
        Xm = make_double2(q2.x - 2 * q1.x, q2.y - 2 * q1.y);
      }
      const double2 e = make_double2(Xk.x + Xm.x, Xk.y - Xm.y);
      const double2 d = make_double2(Xk.x - Xm.x, Xk.y + Xm.y);
      const double2 q = cmul(o[4], d);
      dst[k] = make_double2(e.x - q.y, e.y + q.x);
      if (m != k) dst[m] = make_double2(e.x + q.y, -e.y + q.x);
      if (u == NPP - 1 && ptv == 0) {
        // the DC / Nyquist pair: wave-uniform addresses (scalar loads, requested in the
        // template's first interval), formed with its last batch by one lane
        double xk, xm;
        if (RATIO) {
          const bool second = (tcur & 1) != 0;
          xk = cmul(second ? dc[1] : dc[0], second ? Vc[0] : Sc[0]).x;
          xm = cmul(second ? dc[3] : dc[2], second ? Vc[n2] : Sc[n2]).x;
        } else {
          const double2 a1 = cmul(dc[0], Sc[0]), a2 = cmul(dc[1], Vc[0]);
          const double2 b1 = cmul(dc[2], Sc[n2]), b2 = cmul(dc[3], Vc[n2]);
          xk = a2.x - 2 * a1.x, xm = b2.x - 2 * b1.x;
        }
        // numpy irfft ignores the imaginary parts of the DC and Nyquist bins
        dst[0] = make_double2(xk + xm, xk - xm);
      }
    };
    const int nbar = (P12 && fold) ? 4 : 5;
    issue(0, 0, op[0]);
    if (NPP == 2) issue(0, 1, op[1]);
    for (int it = -1; it < TI; it++) {
      const int tn = it + 1;   // the image this iteration forms
      if (tn < TI) {
        double2 *dst = img(tn);
        {
          const int tt = RATIO ? (tn >> 1) : tn;
          const double2 *Fp = tfft + (int64_t)tt * (n2 + 1);
          const double2 *F2p = tfft2 + (int64_t)tt * (n2 + 1);
          dc[0] = Fp[0], dc[1] = F2p[0], dc[2] = Fp[n2], dc[3] = F2p[n2];
        }
        if (NPP == 2) {
          // two batches, two buffers: each buffer is refilled with the NEXT
          // template's batch as soon as it has been consumed -- a whole template
          // (five intervals) ahead
#pragma unroll
          for (int u = 0; u < 2; u++) {
            form(dst, u, op[u], tn);
            if (tn + 1 < TI) issue(tn + 1, u, op[u]);
            __syncthreads();
          }
        } else {
          // one pair batch per interval, the next one requested first (in flight
          // across the barrier)
#pragma unroll
          for (int u = 0; u < NPP; u++) {
            if (u + 1 < NPP)
              issue(tn, u + 1, op[(u + 1) & 1]);
            else if (tn + 1 < TI)
              issue(tn + 1, 0, op[(u + 1) & 1]);
            form(dst, u, op[u & 1], tn);
            __syncthreads();
          }
        }
        for (int q = NPP; q < nbar; q++) __syncthreads();
      } else {
        for (int q = 0; q < nbar; q++) __syncthreads();
      }
    }
    return;
  }
  // ---- consumers ---------------------------------------------------------
  const int tvx = min(pt, nvel - 1);
  const int pre_pos = lag_pos[min(pt, nlag - 1)];
  const int pre_lo = ilo[tvx];
  const double pre_xg = vgrid[tvx];
  const double pre_x0 = lag_vel[pre_lo], pre_x1 = lag_vel[pre_lo + 1];
  for (int it = -1; it < TI; it++) {
    if (it < 0) {
      const int nbar = (P12 && fold) ? 4 : 5;
      for (int q = 0; q < nbar; q++) __syncthreads();
      continue;
    }
    double2 *a = img(it);
    double *out = chisq + ((int64_t)b * T + (RATIO ? (it >> 1) : it)) * nvel;
    double pre_old = 0;
    if (beta != 0.0) pre_old = out[tvx];
    if (pt < (n2 >> 3)) xw_pass<1, LOG2N>(a, T1, 0, pt);
    __syncthreads();
    if (pt < (n2 >> 3)) xw_pass<1, LOG2N>(a, T1, 1, pt, P12 ? T1c : nullptr);
    __syncthreads();
    if (!P12) {
      // nfft 4096: a third radix-8 pass (M = 32) and the radix-4 pass (M = 4)
      if (pt < (n2 >> 3)) xw_pass<1, LOG2N>(a, T1, 2, pt);
      __syncthreads();
      {
        const int base = pt << 2;
        double2 v0 = a[base], v1 = a[base + 1], v2 = a[base + 2], v3 = a[base + 3];
        dft4<1>(v0, v1, v2, v3);
        a[base] = v0, a[base + 1] = v1, a[base + 2] = v2, a[base + 3] = v3;
      }
      __syncthreads();
    } else if (fold) {
      // the last two passes as one (fft_lds): eight lanes per 64-block
      const int s = pt & 7;
      const int B = pt >> 3;   // n2 / 64 = 64 blocks: one trip
      const unsigned m0 = pmask.b0 & 1u, m63 = pmask.b7 >> 7;
      if (m0 | m63) {
        double2 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = a[B * 64 + s + 8 * j];
        dft8<1>(v);
        double2 w = T1c[64 + s];   // = T1[s << (log2n - 6)]
        w.y = -w.y;
        double2 x63 = cmul(v[7], w), x0 = v[0];
        auto sum8 = [](double x) {
          x += dpp_get<0xb1, 0xf, 0xf>(x);    // quad_perm [1,0,3,2]
          x += dpp_get<0x4e, 0xf, 0xf>(x);    // quad_perm [2,3,0,1]
          x += dpp_get<0x141, 0xf, 0xf>(x);   // row_half_mirror
          return x;
        };
        if (m0) {
          x0.x = sum8(x0.x);
          x0.y = sum8(x0.y);
          if (s == 0) a[B * 64] = x0;
        }
        if (m63) {
          x63.x = sum8(x63.x);
          x63.y = sum8(x63.y);
          if (s == 0) a[B * 64 + 63] = x63;
        }
      }
      __syncthreads();
    } else {
      {   // pass 2 (M = 64, Mp = 8): selected outputs only
        const int blk = pt >> 3, r = pt & 7, base = (blk << 6) + r;
        const unsigned mask = prune[blk];
        if (mask != 0) {
          double2 v[8];
#pragma unroll
          for (int j = 0; j < 8; j++) v[j] = a[base + 8 * j];
          dft8<1>(v);
          double2 wp[8];
          wp[0] = make_double2(1.0, 0.0);
          wp[1] = T1[r << (log2n + 1 - 6 - 1)];
          wp[2] = cmul(wp[1], wp[1]);
          wp[3] = cmul(wp[2], wp[1]);
          wp[4] = cmul(wp[2], wp[2]);
          wp[7] = cmul(wp[4], wp[3]);
          wp[5] = cmul(wp[4], wp[1]);
          wp[6] = cmul(wp[4], wp[2]);
#pragma unroll
          for (int q = 0; q < 8; q++)
            if (mask & (1u << q))
              a[base + 8 * q] = (q == 0) ? v[0] : cmul(v[q], wp[q]);
        }
      }
      __syncthreads();
      {   // pass 3 (M = 8): no twiddles
        const unsigned mask = pmask.g;
        if (mask != 0) {
          const int base = pt << 3;
          double2 v[8];
#pragma unroll
          for (int j = 0; j < 8; j++) v[j] = a[base + j];
          dft8<1>(v);
#pragma unroll
          for (int q = 0; q < 8; q++)
            if (mask & (1u << q)) a[base + q] = v[q];
        }
      }
      __syncthreads();
    }
    if (RATIO) {
      if (pt < nlag) {
        const double c = reinterpret_cast<const double *>(a)[pre_pos] * inv_n;
        if (it & 1)
          c0[pt] = -cA[pt] * cA[pt] / c;
        else
          cA[pt] = c;
      }
    } else if (pt < nlag) {
      c0[pt] = reinterpret_cast<const double *>(a)[pre_pos] * inv_n;
    }
    __syncthreads();
    if ((!RATIO || (it & 1)) && pt < nvel) {
      const double sl = (c0[pre_lo + 1] - c0[pre_lo]) / (pre_x1 - pre_x0);
      const double val = sl * (pre_xg - pre_x0) + c0[pre_lo];
      out[pt] = (beta != 0.0) ? beta * pre_old + val : val;
    }
  }
}

// nfft = 4096 (BASELINE configs[0-1]: one arm of ~2000 px): a radix-8 pass of the
// 2048-point image has 256 butterflies -- half of the consumer lanes -- and lasts as
// long as one of the 4096-point image (one butterfly per lane either way: the pass is
// a latency chain).  Here an iteration takes TWO templates: consumer lane pt works on
// image pt >> 8, the producers form both images (four pair batches, as at nfft 8192).
// Passes 8, 8, 8, 4, nothing pruned; five barriers per iteration.
__global__ void __launch_bounds__(XW_NT)
    ccf_xcorr_ws2_kernel(const double2 *__restrict__ work,
                         const double2 *__restrict__ tfft,
                         const double2 *__restrict__ tfft2, int T,
                         const double2 *__restrict__ tw,
                         const int32_t *__restrict__ lag_pos,
                         const double *__restrict__ lag_vel, int nlag,
                         const int32_t *__restrict__ ilo,
                         const double *__restrict__ vgrid, int nvel, double beta,
                         double *__restrict__ chisq) {
  extern __shared__ double2 fa[];
  constexpr int LOG2N = 11, n2 = 1 << LOG2N, nfft = 2 * n2, npair = n2 / 2;
  constexpr int NBAR = 5;
  // images: iteration parity i, template j of the iteration
  auto img = [&](int i, int j) -> double2 * { return fa + (((i & 1) << 1) + j) * n2; };
  const double2 *T1 = fa + 4 * n2;                         // [n2 / 8]
  double *c0 = reinterpret_cast<double *>(fa + 4 * n2 + XC_NTW(n2));  // [2][nlag]
  const int b = blockIdx.x, tid = threadIdx.x;
  const bool producer = tid >= XW_HALF;
  const int pt = tid & (XW_HALF - 1);
  const double inv_n = 1.0 / nfft;
  xc_fill_twiddles<XW_NT>(fa + 3 * n2, n2, tw);   // -> fa[4 n2 + i]
  const double2 *Sc = work + ((int64_t)b * 2) * (n2 + 1);
  const double2 *Vc = Sc + (n2 + 1);
  typedef int v4i_t __attribute__((ext_vector_type(4)));
  auto ld = [](const __amdgpu_buffer_rsrc_t &r, int off) {
    const v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_double2(__hiloint2double(v.y, v.x), __hiloint2double(v.w, v.z));
  };
  const int nbytes = (n2 + 1) * 16;
  const int NIT = (T + 1) >> 1;
  __syncthreads();
  if (producer) {
    double2 Sk[2], Vk[2], Sm[2], Vm[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int k = 1 + pt + u * XW_HALF, m = n2 - k;
      Sk[u] = Sc[k], Vk[u] = Vc[k], Sm[u] = Sc[m], Vm[u] = Vc[m];
    }
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(
        (void *)tw, 0, (npair + 1) * 16, 0x00020000);
    double2 op[2][5];
    // batch q of an iteration: template 2 i + (q >> 1), pair q & 1
    auto issue = [&](int t, int u, double2 *o) {
      const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc(
          (void *)(tfft + (int64_t)t * (n2 + 1)), 0, nbytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rF2 = __builtin_amdgcn_make_buffer_rsrc(
          (void *)(tfft2 + (int64_t)t * (n2 + 1)), 0, nbytes, 0x00020000);
      const int k = 1 + pt + u * XW_HALF;
      const int ok = k * 16, om = (n2 - k) * 16;
      o[0] = ld(rF, ok), o[1] = ld(rF2, ok);
      o[2] = ld(rF, om), o[3] = ld(rF2, om);
      o[4] = ld(rT, ok);
    };
    auto form = [&](double2 *dst, int t, int u, const double2 *o) {
      const int k = 1 + pt + u * XW_HALF, m = n2 - k;
      const double2 p1 = cmul(o[0], Sk[u]), p2 = cmul(o[1], Vk[u]);
      const double2 Xk = make_double2(p2.x - 2 * p1.x, p2.y - 2 * p1.y);
      const double2 q1 = cmul(o[2], Sm[u]), q2 = cmul(o[3], Vm[u]);
      const double2 Xm = make_double2(q2.x - 2 * q1.x, q2.y - 2 * q1.y);
      const double2 e = make_double2(Xk.x + Xm.x, Xk.y - Xm.y);
      const double2 d = make_double2(Xk.x - Xm.x, Xk.y + Xm.y);
      const double2 q = cmul(o[4], d);
      dst[k] = make_double2(e.x - q.y, e.y + q.x);
      if (m != k) dst[m] = make_double2(e.x + q.y, -e.y + q.x);
      if (u == 1 && pt == 0) {   // DC / Nyquist pair (scalar loads: uniform addresses)
        const double2 *Fp = tfft + (int64_t)t * (n2 + 1);
        const double2 *F2p = tfft2 + (int64_t)t * (n2 + 1);
        const double2 a1 = cmul(Fp[0], Sc[0]), a2 = cmul(F2p[0], Vc[0]);
        const double2 b1 = cmul(Fp[n2], Sc[n2]), b2 = cmul(F2p[n2], Vc[n2]);
        const double xk = a2.x - 2 * a1.x, xm = b2.x - 2 * b1.x;
        dst[0] = make_double2(xk + xm, xk - xm);
      }
    };
    // template of batch q of iteration i, clamped (an odd T forms its last template
    // twice: the copy is never read back)
    auto tq = [&](int i, int q) { return min(2 * i + (q >> 1), T - 1); };
    issue(tq(0, 0), 0, op[0]);
    for (int it = -1; it < NIT; it++) {
      const int in = it + 1;   // the iteration whose images are formed now
      if (in < NIT) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (q + 1 < 4)
            issue(tq(in, q + 1), (q + 1) & 1, op[(q + 1) & 1]);
          else if (in + 1 < NIT)
            issue(tq(in + 1, 0), 0, op[(q + 1) & 1]);
          form(img(in, q >> 1), tq(in, q), q & 1, op[q & 1]);
          __syncthreads();
        }
        __syncthreads();
      } else {
        for (int q = 0; q < NBAR; q++) __syncthreads();
      }
    }
    return;
  }
  // ---- consumers: lane pt on image j = pt >> 8, butterfly bt = pt & 255 ----
  const int j = pt >> 8, bt = pt & 255;
  int pre_pos[1], dummy = 0;
  (void)dummy;
  pre_pos[0] = lag_pos[min(pt, nlag - 1)];
  const int tvx = min(pt, nvel - 1);
  const int pre_lo = ilo[tvx];
  const double pre_xg = vgrid[tvx];
  const double pre_x0 = lag_vel[pre_lo], pre_x1 = lag_vel[pre_lo + 1];
  for (int it = -1; it < NIT; it++) {
    if (it < 0) {
      for (int q = 0; q < NBAR; q++) __syncthreads();
      continue;
    }
    double2 *a = img(it, j);
    const int t0 = 2 * it, t1 = min(2 * it + 1, T - 1);
    const bool two = (2 * it + 1 < T);
    double *out0 = chisq + ((int64_t)b * T + t0) * nvel;
    double *out1 = chisq + ((int64_t)b * T + t1) * nvel;
    double old0 = 0, old1 = 0;
    if (beta != 0.0) {
      old0 = out0[tvx];
      old1 = out1[tvx];
    }
    xw_pass<1, LOG2N>(a, T1, 0, bt);
    __syncthreads();
    xw_pass<1, LOG2N>(a, T1, 1, bt);
    __syncthreads();
    xw_pass<1, LOG2N>(a, T1, 2, bt);
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; h++) {   // radix-4 pass: 512 butterflies per image
      const int base = (bt + 256 * h) << 2;
      double2 v0 = a[base], v1 = a[base + 1], v2 = a[base + 2], v3 = a[base + 3];
      dft4<1>(v0, v1, v2, v3);
      a[base] = v0, a[base + 1] = v1, a[base + 2] = v2, a[base + 3] = v3;
    }
    __syncthreads();
    if (pt < nlag) {
      c0[pt] = reinterpret_cast<const double *>(img(it, 0))[pre_pos[0]] * inv_n;
      c0[nlag + pt] = reinterpret_cast<const double *>(img(it, 1))[pre_pos[0]] * inv_n;
    }
    __syncthreads();
    if (pt < nvel) {
      const double dx = pre_x1 - pre_x0, xg = pre_xg - pre_x0;
      const double sl0 = (c0[pre_lo + 1] - c0[pre_lo]) / dx;
      const double val0 = sl0 * xg + c0[pre_lo];
      out0[pt] = (beta != 0.0) ? beta * old0 + val0 : val0;
      if (two) {
        const double sl1 = (c0[nlag + pre_lo + 1] - c0[nlag + pre_lo]) / dx;
        const double val1 = sl1 * xg + c0[nlag + pre_lo];
        out1[pt] = (beta != 0.0) ? beta * old1 + val1 : val1;
      }
    }
  }
}

extern "C" int rvs_ccf_xcorr(const double *proc_spec, const double *proc_ivar,
                             int nfft, int B, const double *tfft,
                             const double *tfft2, int T, const double *twid_,
                             int continuum, const int32_t *lag_pos,
                             const double *lag_vel, int nlag, const int32_t *ilo,
                             const double *vgrid, int nvel, double beta,
                             const uint8_t *prune, double *chisq, double *work,
                             void *stream) {
  int log2n = 0;
  while ((2 << log2n) < nfft) log2n++;
  if ((2 << log2n) != nfft || nfft < 64 || nfft > 16384) return RVS_E_ARG;
  if (B < 1 || T < 1 || T > 65535 || nlag < 2 || nvel < 1) return RVS_E_ARG;
  const int n2 = nfft >> 1;
  const size_t shm1 = sizeof(double2) * (size_t)(n2 + XC_NTW(n2));
  const size_t shm2 = shm1 + sizeof(double) * 2 * (size_t)nlag;
  if (shm2 > 159 * 1024) return RVS_E_ARG;
  hipStream_t st = rvs_stream(stream);
  const double2 *tw = reinterpret_cast<const double2 *>(twid_);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)ccf_rfft_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              159 * 1024);
    (void)hipFuncSetAttribute((const void *)ccf_xcorr_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              159 * 1024);
    (void)hipGetLastError();
    attr_set = true;
  }
  hipLaunchKernelGGL(ccf_rfft_kernel, dim3(B, 2), dim3(XC_NT), shm1, st,
                     proc_spec, proc_ivar, nfft, log2n, tw,
                     reinterpret_cast<double2 *>(work));
  RVS_LAUNCH_CHECK();
  const int G = xc_group(T, nfft);
  {
    // the wave-specialised persistent form (one block per spectrum), where it applies
    const bool ws_on = rvs_opt(RVS_OPT_XC_WS) != 0;   // xc_ws = 0: the per-pair kernel
    const bool p12 = (nfft == 8192 && prune), p11 = (nfft == 4096);
    if (ws_on && (p12 || p11) && nlag <= XW_HALF && nvel <= XW_HALF && T >= 2) {
      static bool ws_attr = false;
      if (!ws_attr) {
        (void)hipFuncSetAttribute((const void *)ccf_xcorr_ws_kernel<12, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  159 * 1024);
        (void)hipFuncSetAttribute((const void *)ccf_xcorr_ws_kernel<11, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  159 * 1024);
        (void)hipFuncSetAttribute((const void *)ccf_xcorr_ws_kernel<12, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  159 * 1024);
        (void)hipFuncSetAttribute((const void *)ccf_xcorr_ws_kernel<11, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  159 * 1024);
        (void)hipFuncSetAttribute((const void *)ccf_xcorr_ws2_kernel,
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  159 * 1024);
        (void)hipGetLastError();
        ws_attr = true;
      }
      const size_t shmw = sizeof(double2) * (size_t)(2 * n2 + XC_NTW(n2)) +
                          sizeof(double) * (size_t)nlag +
                          sizeof(double2) * (72 + 1) +   // (+ pass 1's twiddles: T1c)
                          (continuum ? 0 : sizeof(double) * (size_t)nlag);   // (cA)
#define RVS_XW_ARGS                                                                 \
  reinterpret_cast<const double2 *>(work), reinterpret_cast<const double2 *>(tfft),  \
      reinterpret_cast<const double2 *>(tfft2), T, tw, lag_pos, lag_vel, nlag, ilo,  \
      vgrid, nvel, beta, prune, chisq
      if (!continuum) {
        // -c0^2 / c1: one correlation per iteration (the two-templates-per-iteration
        // form of nfft 4096 pairs TEMPLATES, so 4096 takes the one-image form here)
        if (p12)
          hipLaunchKernelGGL((ccf_xcorr_ws_kernel<12, true>), dim3(B), dim3(XW_NT), shmw,
                             st, RVS_XW_ARGS);
        else
          hipLaunchKernelGGL((ccf_xcorr_ws_kernel<11, true>), dim3(B), dim3(XW_NT), shmw,
                             st, RVS_XW_ARGS);
        RVS_LAUNCH_CHECK();
        return 0;
      }
      if (p12)
        hipLaunchKernelGGL((ccf_xcorr_ws_kernel<12, false>), dim3(B), dim3(XW_NT), shmw, st,
                           reinterpret_cast<const double2 *>(work),
                           reinterpret_cast<const double2 *>(tfft),
                           reinterpret_cast<const double2 *>(tfft2), T, tw, lag_pos,
                           lag_vel, nlag, ilo, vgrid, nvel, beta, prune, chisq);
      else if (rvs_opt(RVS_OPT_XC_WS1))   // (one template per iteration: measured 1.10
                                          // ms per 1000 spectra against 0.84; per pair 1.34)
        hipLaunchKernelGGL((ccf_xcorr_ws_kernel<11, false>), dim3(B), dim3(XW_NT), shmw, st,
                           reinterpret_cast<const double2 *>(work),
                           reinterpret_cast<const double2 *>(tfft),
                           reinterpret_cast<const double2 *>(tfft2), T, tw, lag_pos,
                           lag_vel, nlag, ilo, vgrid, nvel, beta, prune, chisq);
      else
        hipLaunchKernelGGL(ccf_xcorr_ws2_kernel, dim3(B), dim3(XW_NT),
                           sizeof(double2) * (size_t)(4 * n2 + XC_NTW(n2)) +
                               sizeof(double) * 2 * (size_t)nlag,
                           st, reinterpret_cast<const double2 *>(work),
                           reinterpret_cast<const double2 *>(tfft),
                           reinterpret_cast<const double2 *>(tfft2), T, tw, lag_pos,
                           lag_vel, nlag, ilo, vgrid, nvel, beta, chisq);
      RVS_LAUNCH_CHECK();
      return 0;
    }
  }
  const int bmax = G ? (int)(0x7fffffffll / xc_nblocks(T, 1, G)) : 65535;
  for (int b0 = 0; b0 < B; b0 += bmax) {
    const int nb = (B - b0 < bmax) ? (B - b0) : bmax;
    const dim3 grid = G ? dim3((unsigned)xc_nblocks(T, nb, G)) : dim3(T, nb);
    hipLaunchKernelGGL(
        ccf_xcorr_kernel, grid, dim3(XB_NT), shm2, st,
        reinterpret_cast<const double2 *>(work) + (int64_t)b0 * 2 * (n2 + 1),
        nfft, log2n, reinterpret_cast<const double2 *>(tfft),
        reinterpret_cast<const double2 *>(tfft2), T, tw, continuum, lag_pos,
        lag_vel, nlag, ilo, vgrid, nvel, beta, prune,
        chisq + (int64_t)b0 * T * nvel, nb, G);
    RVS_LAUNCH_CHECK();
  }
  return 0;
}
