// EXPERIMENT (round 4; built, verified to 4e-16 against the shipped kernel, measured
// SLOWER: 46.7 us per (job, arm) against 40.3 -- DESIGN 4.7).  Not part of
// librvsgpu.so; tools/perf/obj_bench.hip builds it with -DOBJ_PIPE_EXPERIMENT.
//
// objective_pipe.hip -- the optimiser's objective (get_chisq at one point per
// job, spec_fit.py:797-989; chisq_func of vel_fit.py:205-254) as a PERSISTENT
// kernel with wave specialisation.
//
// objective.hip's kernel gives a (job, arm) to a 512-thread block that keeps the
// whole template in LDS, so a CU holds one block and its phases run back to back:
// the vertex-row gather (2^ndim float32 rows per item, 400 KB out of the Infinity
// Cache / HBM, 27 % of the block's time) with the VALU idle, then FIR, spline
// solve, model, normal equations, Cholesky, residuals with the memory pipe idle.
// Here one 768-thread block per CU walks a list of items:
//   waves 0-7  (consumers)  FIR -> spline solve -> model -> chi^2 of item n out of
//                           LDS, the arithmetic of objective.hip phase by phase;
//   waves 8-11 (producers)  gather + blend + exp of item n+2 INTO REGISTERS
//                           (<= 32 doubles per thread), a group of rows requested
//                           at one barrier and consumed at a later one, so that
//                           the round trips lie under the consumers' phases; the
//                           finished template of item n+1 goes into LDS at the
//                           point of item n where the buffer is free (after the
//                           model pass), the rotational kernel of item n+1 is
//                           built by producer wave 8 beside the chi^2 of item n.
// gfx950 has one barrier per workgroup: both roles execute the same NUMBER of
// s_barrier per item (PIPE_NBAR), each role in its own loop (so that the register
// allocator sees two programs, not one with both roles' live ranges).  The
// barrier's fence does not wait for vector loads (workgroup scope, no
// threadgroup split), so the producers' requests stay in flight across it.
// 168 VGPRs (three waves per SIMD): the P(P+3)/2 normal-equation sums are
// accumulated in two passes over the pixels.
#include "../../../rvspecfit_amd/csrc/objective_dev.h"

#define PIPE_NT 768
#define PIPE_NC 512            // consumer threads (waves 0..7)
#define PIPE_NCW 8
#define PIPE_NP 256            // producer threads (waves 8..11)
#define PIPE_NPW 4
#define PIPE_NBAR 13           // barriers per item, both roles
#define PIPE_GPX (4 * PIPE_NP) // template points per gather group
#define PIPE_NG 8              // groups: ntp <= 8192
#define PIPE_CHMAX 16          // rows of a consumer's Thomas chunk (8192 / 512)

#define PIPE_BAR() __syncthreads()
#ifndef PIPE_DBG_SKIP
#define PIPE_DBG_SKIP 0   // (compile-time experiments: phases left out)
#endif

// what the producers leave for the consumers of an item (and, in `w`/`id`, what
// objective_locate_kernel left for the producers)
struct PipeHdr {
  double w[OBJ_LOC_NV];
  int64_t id[OBJ_LOC_NV];
  double dist;           // outside flag (kd distance / the evaluator's flag)
  double pmax[PIPE_NPW], pnan[PIPE_NPW];   // MAX_VAL guard partials
  int mode, nearest;
  int kmax, copy, st_extra;
};

// rows [0, PA) of the normal equations go into the first accumulation pass:
// the smallest PA with PA (PA + 3) / 2 >= half of the P (P + 3) / 2 sums
__host__ __device__ constexpr int pipe_split(int P) {
  int pa = 1;
  while (pa < P && pa * (pa + 3) < P * (P + 3) / 2) pa++;
  return P <= 6 ? P : pa;
}

template <int P, bool FROMT>
__global__ void __launch_bounds__(PIPE_NT)
    objective_pipe_kernel(ObjArms A, ObjTempl TT, const double *__restrict__ locrec,
                          const double *__restrict__ vsini,
                          const int32_t *__restrict__ job_spec, int J, int nmax,
                          const double *__restrict__ vel, double eps_ld,
                          double *__restrict__ armchi, int32_t *__restrict__ armst,
                          double *__restrict__ armout) {
  constexpr int NT = P * (P + 1) / 2;
  constexpr int NV = NT + P;
  extern __shared__ double lds[];
  __shared__ PipeHdr H[2];
  __shared__ double red[PIPE_NCW][NV + 1];
  __shared__ double coefs[P + 2];
  __shared__ double Lm[P][P + 1];
  __shared__ double ldv[P];
  __shared__ double red8[PIPE_NCW];
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int tid_k = tid, lane_k = lane, w_k = w;
  double *bufA = lds, *bufB = lds + nmax, *bufC = lds + 2 * (size_t)nmax;
  const int total = J * A.n;
  // items of this block: q = blockIdx.x + n * gridDim.x, arm-major
  const int cnt = (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

  if (w >= PIPE_NCW) {
    // =====================================================================
    // producers
    // =====================================================================
#ifdef PIPE_DBG_NOPROD
    for (int n = -2; n < cnt; n++)
      for (int b = 0; b < PIPE_NBAR; b++) PIPE_BAR();
    return;
#endif
    const int pt = tid - PIPE_NC, pw = w - PIPE_NCW;
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
    double tr[4 * PIPE_NG];   // the template being gathered (this thread's points)
    f4u rn[16];               // vertex rows in flight
    d2u rd0 = {0, 0}, rd1 = {0, 0};   // FROMT: the row itself
    double gmx = 0;           // MAX_VAL guard of the item being gathered
    bool gnan = false;
#pragma unroll
    for (int i = 0; i < 4 * PIPE_NG; i++) tr[i] = 0;
#pragma unroll
    for (int u = 0; u < 16; u++) rn[u] = f4u{0, 0, 0, 0};

    // the item being gathered (n + 2 in the steady state)
    int g_arm = 0, g_j = 0, g_N = 0, g_mode = 0;
    bool g_valid = false;
    // ... and the one after it (its cell record is fetched ahead)
    int n_arm = 0, n_j = 0, n_N = 0, n_mode = 0;
    bool n_valid = false;
    auto n_setup = [&](int m) {   // uniform
      n_valid = (m >= 0 && m < cnt);
      n_mode = 0;
      if (n_valid) {
        const int q = blockIdx.x + m * gridDim.x;
        n_arm = q / J;
        n_j = q - n_arm * J;
        n_N = A.a[n_arm].ntp;
      }
    };
    // request group g of the item being gathered (H[hp] holds its cell record)
    auto issue = [&](int g, const PipeHdr &Hd) {
      if (!g_valid) return;
      const int k = g * PIPE_GPX + 4 * pt;
      const int N = g_N;
      if (k >= (N & ~3)) return;
      if (FROMT) {
        const double *row = TT.templ[g_arm] + (int64_t)g_j * N + k;
        rd0 = *reinterpret_cast<const d2u *>(row);
        rd1 = *reinterpret_cast<const d2u *>(row + 2);
      } else if (g_mode == 0) {
        const rvs_objective_arm &T = A.a[g_arm];
        const int nv = 1 << T.ndim;
#pragma unroll
        for (int u = 0; u < 16; u++)
          rn[u] = *reinterpret_cast<const f4u *>(T.dats + Hd.id[min(u, nv - 1)] * N + k);
      } else {
        const rvs_objective_arm &T = A.a[g_arm];
        rn[0] = *reinterpret_cast<const f4u *>(T.dats + (int64_t)Hd.nearest * N + k);
      }
    };
    // blend + exp of group g into tr[4g .. 4g+3]
    auto consume = [&](int g, const PipeHdr &Hd, double *t4) {
      if (!g_valid) return;
      const int k = g * PIPE_GPX + 4 * pt;
      const int N = g_N;
      if (k >= (N & ~3)) return;
      if (FROMT) {
        t4[0] = rd0.x, t4[1] = rd0.y, t4[2] = rd1.x, t4[3] = rd1.y;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (!(t4[q] == t4[q])) gnan = true;
          gmx = fmax(gmx, fabs(t4[q]));
        }
      } else if (g_mode == 0) {
        const rvs_objective_arm &T = A.a[g_arm];
        const int nv = 1 << T.ndim;
        double a4[4] = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 16; u++) {
          if (u < nv) {
            const double wv = Hd.w[u];
            a4[0] = fma(wv, (double)rn[u].x, a4[0]);
            a4[1] = fma(wv, (double)rn[u].y, a4[1]);
            a4[2] = fma(wv, (double)rn[u].z, a4[2]);
            a4[3] = fma(wv, (double)rn[u].w, a4[3]);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) t4[q] = T.exp_flag ? exp(a4[q]) : a4[q];
      } else {
        const rvs_objective_arm &T = A.a[g_arm];
        const float r4[4] = {rn[0].x, rn[0].y, rn[0].z, rn[0].w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const double val = T.exp_flag ? (double)np_expf(r4[q]) : (double)r4[q];
          t4[q] = val;
          if (!(val == val)) gnan = true;
          gmx = fmax(gmx, fabs(val));
        }
      }
    };
#define P_STEP(gc, gi, hp)                              \
  do {                                                  \
    if ((gc) >= 0) consume((gc), H[hp], &tr[4 * (gc)]); \
    if ((gi) >= 0 && (gi) < PIPE_NG) issue((gi), H[hp]); \
  } while (0)

    // periods -2, -1 fill the pipe (the consumers only meet the barriers)
    for (int n = -2; n < cnt; n++) {
      const int hp = n & 1;        // H[hp]: cell record of the item being gathered
                                   //        in slots 1-7 that is item n+1: H[(n+1)&1]
      const int hq = (n + 1) & 1;
      // ---- slots 1..8: groups 3..6 of item n+1 (a group is consumed two or more
      // consumer phases after its request: ~2.5 us, the round trip under load)
      P_STEP(2, 3, hq);
      PIPE_BAR();   // 1
      PIPE_BAR();   // 2
      P_STEP(3, 4, hq);
      PIPE_BAR();   // 3
      // cell record of item n+2 -> H[hp] (the consumers read H[hp] in slot 1 only)
      n_setup(n + 2);
      if (n_valid) {
        if (!FROMT) {
          const double *r = locrec + ((int64_t)n_arm * J + n_j) * OBJ_LOC_REC;
          if (pt < OBJ_LOC_NV) {
            H[hp].w[pt] = r[pt];
            H[hp].id[pt] = reinterpret_cast<const int64_t *>(r)[OBJ_LOC_NV + pt];
          }
          const int32_t *mi = reinterpret_cast<const int32_t *>(r + 2 * OBJ_LOC_NV + 1);
          n_mode = mi[0];   // (every producer: its branch in issue / consume)
          if (pt == 64) {
            H[hp].dist = r[2 * OBJ_LOC_NV];
            H[hp].mode = mi[0];
            H[hp].nearest = mi[1];
          }
        } else {
          const double o = TT.outside[n_arm][n_j];
          n_mode = (o == 0.0) ? 0 : 1;
          if (pt == 64) {
            H[hp].dist = o;
            H[hp].mode = n_mode;
            H[hp].nearest = 0;
          }
        }
      }
      PIPE_BAR();   // 4
      PIPE_BAR();   // 5
      P_STEP(4, 5, hq);
      PIPE_BAR();   // 6
      PIPE_BAR();   // 7
      P_STEP(5, 6, hq);
      PIPE_BAR();   // 8
      // ---- slot 9: hand-over of item n+1 -----------------------------------------
      P_STEP(6, -1, hq);
      if (g_valid && g_N > 7 * PIPE_GPX) {   // (ntp > 7168: an eighth group)
        issue(7, H[hq]);
        consume(7, H[hq], &tr[28]);
      }
      if (g_valid) {
        const int N = g_N;
        const int N4 = N & ~3;
#pragma unroll
        for (int g = 0; g < PIPE_NG; g++) {
          const int k = g * PIPE_GPX + 4 * pt;
          if (k < N4) {
#pragma unroll
            for (int q = 0; q < 4; q++) bufA[k + q] = tr[4 * g + q];
          }
        }
        // the last N % 4 points (rows start on 4-byte boundaries only)
        if (pt < N - N4) {
          const int k = N4 + pt;
          double val;
          if (FROMT) {
            val = TT.templ[g_arm][(int64_t)g_j * N + k];
          } else if (g_mode == 0) {
            const rvs_objective_arm &T = A.a[g_arm];
            const int nv = 1 << T.ndim;
            double acc = 0;
            for (int v = 0; v < nv; v++)
              acc = fma(H[hq].w[v], (double)T.dats[H[hq].id[v] * N + k], acc);
            val = T.exp_flag ? exp(acc) : acc;
          } else {
            const rvs_objective_arm &T = A.a[g_arm];
            const float rv = T.dats[(int64_t)H[hq].nearest * N + k];
            val = T.exp_flag ? (double)np_expf(rv) : (double)rv;
          }
          if (FROMT || g_mode != 0) {
            if (!(val == val)) gnan = true;
            gmx = fmax(gmx, fabs(val));
          }
          bufA[k] = val;
        }
        const double mxw = wave_max(gmx);
        const double nnw = wave_sum(gnan ? 1.0 : 0.0);
        if (lane == 0) {
          H[hq].pmax[pw] = mxw;
          H[hq].pnan[pw] = nnw;
        }
        // rotational kernel of item n+1 (wave 8; scratch: bufB, dead since the
        // model pass of item n); left in bufB behind the primitives and moved
        // into bufC in slot 12
        if (pw == 0) {
          int st_extra = 0, kmax = 0;
          bool copy = true;
          if (vsini) {
            const double vs = vsini[g_j];
            const double R = (vs / RVS_C_KMS) / A.a[g_arm].lnstep;
            copy = !(vs > 0) || (R < 1e-9);
            if (!copy) {
              kmax = (int)ceil(R + 1);
              if (kmax >= N || 3 * (kmax + 3) > N) {
                st_extra = RVS_ST_NONFINITE;
                copy = true;
              }
            }
            if (!copy) {
              double *pk0 = bufB, *pk1 = bufB + (kmax + 3), *tp = bufB + 2 * (kmax + 3);
              for (int jj = lane; jj <= kmax + 2; jj += 64) {
                const double x = fmin(fmax((jj - 1) / R, -1.0), 1.0);
                double k0, k1;
                rot_prim(x, eps_ld, k0, k1);
                pk0[jj] = k0;
                pk1[jj] = k1;
              }
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
              double psum = 0;
              for (int k = lane; k <= kmax; k += 64) {
                double ww = 0;
                double lo = fmin(fmax(k / R, -1.0), 1.0),
                       hi = fmin(fmax((k + 1) / R, -1.0), 1.0);
                if (hi > lo)
                  ww += -R * (pk1[k + 2] - pk1[k + 1]) +
                        (1.0 + k) * (pk0[k + 2] - pk0[k + 1]);
                lo = fmin(fmax((k - 1) / R, -1.0), 1.0);
                hi = fmin(fmax(k / R, -1.0), 1.0);
                if (hi > lo)
                  ww += R * (pk1[k + 1] - pk1[k]) + (1.0 - k) * (pk0[k + 1] - pk0[k]);
                tp[k] = ww;
                psum += (k == 0) ? ww : 2 * ww;
              }
              psum = wave_sum(psum);
              const double inv = 1.0 / psum;
              for (int k = lane; k <= kmax; k += 64) tp[k] = tp[k] * inv;
            }
          }
          if (lane == 0) {
            H[hq].kmax = kmax;
            H[hq].copy = copy ? 1 : 0;
            H[hq].st_extra = st_extra;
          }
        }
      }
      // item n+2 becomes the item being gathered; its first group goes out now
      g_valid = n_valid, g_arm = n_arm, g_j = n_j, g_N = n_N, g_mode = n_mode;
      gmx = 0;
      gnan = false;
      P_STEP(-1, 0, hp);
      PIPE_BAR();   // 9
      PIPE_BAR();   // 10
      P_STEP(0, 1, hp);
      PIPE_BAR();   // 11
      P_STEP(1, 2, hp);
      PIPE_BAR();   // 12
      // ---- slot 13: taps of item n+1 into bufC (tcache of item n is dead) -------
      if (pw == 0 && n + 1 >= 0 && n + 1 < cnt && !H[hq].copy) {
        const int kmax = H[hq].kmax;
        const double *tp = bufB + 2 * (kmax + 3);
        for (int k = lane; k <= kmax; k += 64) bufC[k] = tp[k];
      }
      PIPE_BAR();   // 13
    }
#undef P_STEP
    return;
  }

  // =======================================================================
  // consumers
  // =======================================================================
  for (int n = -2; n < cnt; n++) {
#ifdef PIPE_DBG_NOCONS
    if (true) {
#else
    if (n < 0) {
#endif
#pragma unroll
      for (int b = 0; b < PIPE_NBAR; b++) PIPE_BAR();
      continue;
    }
    // (opaque copies: nothing derived from the thread index is hoisted out of
    // the item loop to sit in registers across all of its phases)
    int tid = tid_k, lane = lane_k, w = w_k;
    asm volatile("" : "+v"(tid), "+v"(lane), "+v"(w));
    const int q = blockIdx.x + n * gridDim.x;
    const int arm = q / J;
    const int j = q - arm * J;
    const rvs_objective_arm &T = A.a[arm];
    const int N = T.ntp, m = N - 2;
    // ---- slot 1: header, FIR ------------------------------------------------------
    const PipeHdr &Hd = H[n & 1];
    const int mode = Hd.mode;
    const int kmax = Hd.kmax;
    const bool copy = Hd.copy != 0;
    const int st_extra = Hd.st_extra;
    double outside = 0.0;
    if (mode != 0) {   // MAX_VAL guard of getCurTempl (spec_fit.py:392-397)
      double mm = 0, nn = 0;
#pragma unroll
      for (int i = 0; i < PIPE_NPW; i++) {
        mm = fmax(mm, Hd.pmax[i]);
        nn += Hd.pnan[i];
      }
      outside = Hd.dist;
      if (outside > 0 && (mm > 1e100 || nn > 0 || isinf(mm)))
        outside = __builtin_nan("");
    }
    const bool usable = fabs(outside) <= 1.79e308;
    double *y = bufA, *dp = bufB;
    if (!(PIPE_DBG_SKIP & 1) && vsini && !copy) {
      const int Lc = (N + PIPE_NC - 1) / PIPE_NC;
      const int c0 = tid * Lc, c1 = min(N, c0 + Lc);
      auto in = [&](int qq) { return (qq >= 0 && qq < N) ? bufA[qq] : 0.0; };
      auto tp = [&](int mm) { return bufC[mm < 0 ? -mm : mm]; };
      for (int i0 = c0; i0 < c1; i0 += 4) {
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int qq = i0 - kmax;
        double a0 = in(qq), a1 = in(qq + 1), a2 = in(qq + 2), a3 = in(qq + 3);
        for (int mm = -kmax; mm <= kmax; mm += 4, qq += 4) {
          double t = tp(mm);
          s0 = fma(a0, t, s0), s1 = fma(a1, t, s1);
          s2 = fma(a2, t, s2), s3 = fma(a3, t, s3);
          a0 = in(qq + 4);
          t = (mm + 1 <= kmax) ? tp(mm + 1) : 0.0;
          s0 = fma(a1, t, s0), s1 = fma(a2, t, s1);
          s2 = fma(a3, t, s2), s3 = fma(a0, t, s3);
          a1 = in(qq + 5);
          t = (mm + 2 <= kmax) ? tp(mm + 2) : 0.0;
          s0 = fma(a2, t, s0), s1 = fma(a3, t, s1);
          s2 = fma(a0, t, s2), s3 = fma(a1, t, s3);
          a2 = in(qq + 6);
          t = (mm + 3 <= kmax) ? tp(mm + 3) : 0.0;
          s0 = fma(a3, t, s0), s1 = fma(a0, t, s1);
          s2 = fma(a1, t, s2), s3 = fma(a2, t, s3);
          a3 = in(qq + 7);
        }
        bufB[i0] = s0;
        if (i0 + 1 < c1) bufB[i0 + 1] = s1;
        if (i0 + 2 < c1) bufB[i0 + 2] = s2;
        if (i0 + 3 < c1) bufB[i0 + 3] = s3;
      }
      y = bufB;
      dp = bufA;
    }
    PIPE_BAR();   // 1
    // ---- slot 2: right-hand sides of the spline system ------------------------------
    double *ec = bufC;
    const double *g = T.factors, *e = T.factors + N, *cc = T.factors + 2 * N,
                 *hh = T.factors + 3 * N, *ih = T.factors + 4 * N;
    if (!(PIPE_DBG_SKIP & 2))
    for (int i0 = tid; i0 < m; i0 += 4 * PIPE_NC) {
      double f0[4], f1[4], fg[4], fe[4];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int i = min(i0 + c * PIPE_NC, m - 1);
        f0[c] = ih[i];
        f1[c] = ih[i + 1];
        fg[c] = g[i];
        fe[c] = e[i];
      }
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int i = i0 + c * PIPE_NC;
        if (i < m) {
          const double y1 = y[i + 1];
          const double s0 = (y1 - y[i]) * f0[c], s1 = (y[i + 2] - y1) * f1[c];
          dp[i] = 6 * (s1 - s0) * fg[c];
          ec[i] = fe[c];
        }
      }
    }
    PIPE_BAR();   // 2
    // ---- slots 3-6: chunked Thomas with chunk transfer coefficients -----------------
    const int CH = max(12, (m + PIPE_NC - 1) / PIPE_NC);
    const int a0 = min(m, tid * CH), a1 = min(m, a0 + CH);
    // value entering a chunk from dir = -1 (lower threads) / +1 (upper): the three
    // nearest chunks' coefficients; across a wave boundary through red[]
    auto chain_pub = [&](double al, double be, int dir) {
      const int edge = (dir < 0) ? (63 - lane) : lane;
      if (edge < 3) {
        red[w][2 * edge] = al;
        red[w][2 * edge + 1] = be;
      }
    };
    auto chain_get = [&](double al, double be, int dir) -> double {
      double av3[3], bv3[3];
#pragma unroll
      for (int k = 1; k <= 3; k++) {
        av3[k - 1] = (dir < 0) ? __shfl_up(al, k, 64) : __shfl_down(al, k, 64);
        bv3[k - 1] = (dir < 0) ? __shfl_up(be, k, 64) : __shfl_down(be, k, 64);
      }
      const int mine = (dir < 0) ? lane : (63 - lane);
#pragma unroll
      for (int k = 1; k <= 3; k++) {
        if (mine < k) {
          const int ww = w + dir;
          const int sl = k - 1 - mine;
          const bool have = (ww >= 0 && ww < PIPE_NCW);
          av3[k - 1] = have ? red[ww][2 * sl] : 0.0;
          bv3[k - 1] = have ? red[ww][2 * sl + 1] : 0.0;
        }
      }
      return av3[0] + bv3[0] * (av3[1] + bv3[1] * av3[2]);
    };
    // Both sweeps run twice over a thread's rows: the first time from zero for the
    // chunk's transfer coefficients (d_last, P_last) only, the second time from
    // the value that really enters the chunk.  (objective.hip keeps the first
    // pass's rows and running products in registers, 64 of them, and corrects;
    // re-reading 3 x 13 LDS rows instead is what lets this kernel live in the
    // 168 registers of three waves per SIMD.)
    double loc[PIPE_CHMAX];
    double cd = 0, cpb = 0;
    if (!(PIPE_DBG_SKIP & 2)) {
      double d = 0, pb = 1;
#pragma unroll
      for (int qq = 0; qq < PIPE_CHMAX; qq++)
        if (a0 + qq < a1) {
          const double ei = ec[a0 + qq];
          d = dp[a0 + qq] - ei * d;
          pb = -ei * pb;
        }
      cd = d;
      cpb = pb;
      chain_pub(d, pb, -1);
    }
    PIPE_BAR();   // 3
    if (!(PIPE_DBG_SKIP & 2)) {
      double d = chain_get(cd, cpb, -1);
#pragma unroll
      for (int qq = 0; qq < PIPE_CHMAX; qq++)
        if (a0 + qq < a1) {
          d = dp[a0 + qq] - ec[a0 + qq] * d;
          loc[qq] = d;   // d of the forward sweep
        }
    }
    PIPE_BAR();   // 4: every chunk has read its e
    if (!(PIPE_DBG_SKIP & 2)) {
      for (int i0 = tid; i0 < m; i0 += 4 * PIPE_NC) {
        double fc[4];
#pragma unroll
        for (int c = 0; c < 4; c++) fc[c] = cc[min(i0 + c * PIPE_NC, m - 1)];
#pragma unroll
        for (int c = 0; c < 4; c++)
          if (i0 + c * PIPE_NC < m) ec[i0 + c * PIPE_NC] = fc[c];
      }
    }
    PIPE_BAR();   // 5
    if (!(PIPE_DBG_SKIP & 2)) {
      double z = 0, pb = 1;
#pragma unroll
      for (int qq = PIPE_CHMAX - 1; qq >= 0; qq--)
        if (a0 + qq < a1) {
          const double ci = ec[a0 + qq];
          z = loc[qq] - ci * z;
          pb = -ci * pb;
        }
      cd = z;
      cpb = pb;
      chain_pub(z, pb, +1);
    }
    PIPE_BAR();   // 6
    if (!(PIPE_DBG_SKIP & 2)) {
      double z = chain_get(cd, cpb, +1);
#pragma unroll
      for (int qq = PIPE_CHMAX - 1; qq >= 0; qq--)
        if (a0 + qq < a1) {
          z = loc[qq] - ec[a0 + qq] * z;
          dp[a0 + qq] = z;
        }
    }
    PIPE_BAR();   // 7
#ifndef PIPE_NO_FENCE7
    __builtin_amdgcn_sched_barrier(0);   // (the model pass's loads stay below the solve)
#endif
    // ---- slot 8: model and data in units of sigma -> tcache (bufC) -------------------
    const rvs_point_arm &S = T.pt;
    const int npix = S.npix;
    const int s = job_spec ? job_spec[j] : j;
    const double bb = vel[j] / RVS_C_KMS;
    const double f = sqrt((1.0 - bb) / (1.0 + bb));
    const double espec_sys = S.espec_sys;
    const double sys2 = espec_sys * espec_sys;
    const double *sp = S.spec + (int64_t)s * npix;
    const double *es = S.espec + (int64_t)s * npix;
    const double x0 = S.knots[0], xlast = S.knots[N - 1];
    const double shift = S.log_step ? log(f) / log(S.knots[1] / x0) : 0.0;
    const double lin_inv_step = S.log_step ? 0.0 : 1.0 / (S.knots[1] - x0);
    double *tcache = bufC;
    if (!(PIPE_DBG_SKIP & 4)) {
      constexpr int U = 6;
      for (int kb = tid; kb < npix; kb += U * PIPE_NC) {
        double lm[U], wk[U], e_[U], s_[U], kn[U], hk[U], ik[U];
        int ps[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int k = min(kb + u * PIPE_NC, npix - 1);
          lm[u] = S.lam[k];
          wk[u] = S.log_step ? S.work[k] : 0.0;
          e_[u] = es[k];
          s_[u] = sp[k];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          lm[u] *= f;
          int pos = S.log_step ? (int)(wk[u] + shift)
                               : (int)((lm[u] - x0) * lin_inv_step);
          pos = min(max(pos, 0), N - 2);
          ps[u] = pos;
          kn[u] = S.knots[pos];
          hk[u] = hh[pos];
          ik[u] = ih[pos];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int k = kb + u * PIPE_NC, pos = ps[u];
          const double dl = lm[u] - kn[u];
          const double h = hk[u], hinv = ik[u];
          const double zi = (pos == 0) ? 0.0 : dp[pos - 1];
          const double zi1 = (pos + 1 == N - 1) ? 0.0 : dp[pos];
          const double yi = y[pos], yi1 = y[pos + 1];
          const double t1 = hinv * (1.0 / 6), t2 = h * (1.0 / 6);
          const double cb = (yi1 - yi) * hinv - t2 * (2 * zi + zi1);
          const double c2 = 0.5 * zi, c3 = (zi1 - zi) * t1;
          const double tv = fma(fma(fma(c3, dl, c2), dl, cb), dl, yi);
          double ee = e_[u];
          if (espec_sys > 0) ee = sqrt(sys2 + ee * ee);
          const double ie = 1.0 / ee;
          if (k < npix) {
            tcache[k] = tv * ie;
            tcache[npix + k] = s_[u] * ie;
          }
        }
      }
    }
    PIPE_BAR();   // 8: y, dp are dead -- the producers fill bufA with item n+1
    // ---- slot 9: normal equations, two passes, wave totals into red[w][] -------------
    if (!(PIPE_DBG_SKIP & 8)) {
      constexpr int PA = pipe_split(P);
      constexpr int TA = PA * (PA + 1) / 2;          // matrix sums of pass A
      constexpr int CA = TA + PA;                    // all sums of pass A
      constexpr int CB = NV - CA;
      {
        double vals[CA];
#pragma unroll
        for (int i = 0; i < CA; i++) vals[i] = 0;
        for (int k = tid; k < npix; k += PIPE_NC) {
          const double te = tcache[k], dk = tcache[npix + k];
          const double wt = te * te, u = te * dk;
          const double *prow = S.polysT + (int64_t)k * P;
          double pv[PA];
#pragma unroll
          for (int i = 0; i < PA; i++) pv[i] = prow[i];
#pragma unroll
          for (int i = 0; i < PA; i++) {
            vals[TA + i] = fma(pv[i], u, vals[TA + i]);
#pragma unroll
            for (int jj = 0; jj <= i; jj++)
              vals[TRI(i, jj)] = fma(pv[i], pv[jj] * wt, vals[TRI(i, jj)]);
          }
        }
        int c2, base, lim;
        wave_halve<CA>(vals, lane, c2, base, lim);
#pragma unroll
        for (int i = 0; i < (CA + 63) / 64 + 1; i++)
          if (i < c2 && base + i < lim) {
            const int a = base + i;
            red[w][a < TA ? a : NT + (a - TA)] = vals[i];
          }
      }
      if (CB > 0) {
        double vals[CB > 0 ? CB : 1];
#pragma unroll
        for (int i = 0; i < CB; i++) vals[i] = 0;
        for (int k = tid; k < npix; k += PIPE_NC) {
          const double te = tcache[k], dk = tcache[npix + k];
          const double wt = te * te, u = te * dk;
          const double *prow = S.polysT + (int64_t)k * P;
          double pv[P];
#pragma unroll
          for (int i = 0; i < P; i++) pv[i] = prow[i];
#pragma unroll
          for (int i = PA; i < P; i++) {
            vals[(NT - TA) + (i - PA)] = fma(pv[i], u, vals[(NT - TA) + (i - PA)]);
#pragma unroll
            for (int jj = 0; jj <= i; jj++)
              vals[TRI(i, jj) - TA] = fma(pv[i], pv[jj] * wt, vals[TRI(i, jj) - TA]);
          }
        }
        int c2, base, lim;
        wave_halve<(CB > 0 ? CB : 1)>(vals, lane, c2, base, lim);
#pragma unroll
        for (int i = 0; i < (CB + 63) / 64 + 1; i++)
          if (i < c2 && base + i < lim) {
            const int b = base + i;
            red[w][b < NT - TA ? TA + b : NT + PA + (b - (NT - TA))] = vals[i];
          }
      }
    }
    PIPE_BAR();   // 13
    // ---- slot 10: fold of the waves' totals (wave order) ------------------------------
    if (tid < NV) {
      double v = red[0][tid];
#pragma unroll
      for (int qq = 1; qq < PIPE_NCW; qq++) v += red[qq][tid];
      red[0][tid] = v;
    }
    PIPE_BAR();   // 9
    // ---- slot 11: Cholesky + triangular solves, row i on lane i (objective.hip) ------
    if (!(PIPE_DBG_SKIP & 16) && w == 0) {
      const int i = lane < P ? lane : P - 1;
      double row[P];
#pragma unroll
      for (int jj = 0; jj < P; jj++) row[jj] = (jj <= i) ? red[0][TRI(i, jj)] : 0.0;
      const double vi = red[0][NT + i];
      bool ok = true;
      auto bcast = [](double v, int src) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
        return __hiloint2double(hi, lo);
      };
      double dg = 1.0, rdg = 1.0;
#pragma unroll
      for (int jj = 0; jj < P; jj++) {
        double sum = row[jj];
#pragma unroll
        for (int qq = 0; qq < jj; qq++) sum -= row[qq] * bcast(row[qq], jj);
        double rd = __builtin_amdgcn_rsq(sum);
        {
          const double hx = 0.5 * sum;
          rd = fma(rd, fma(-hx * rd, rd, 0.5), rd);
          rd = fma(rd, fma(-hx * rd, rd, 0.5), rd);
        }
        const double d = sum * rd;
        if (lane == jj) {
          if (!(sum > 0)) ok = false;
          dg = d;
          rdg = rd;
        }
        const double rdj = bcast(rd, jj);
        row[jj] = (lane == jj) ? d : sum * rdj;
        Lm[i][jj] = row[jj];
      }
      if (lane < P) ldv[lane] = log(dg);
      double si = vi;
#pragma unroll
      for (int qq = 0; qq < P; qq++) {
        const double yq = bcast(si * rdg, qq);
        if (i > qq) si -= row[qq] * yq;
        if (lane == qq) si = yq;
      }
      __builtin_amdgcn_wave_barrier();
      double col[P];
#pragma unroll
      for (int ii = 0; ii < P; ii++) col[ii] = Lm[ii][i];
      double ti = si;
#pragma unroll
      for (int ii = P - 1; ii >= 0; ii--) {
        const double aii = bcast(ti * rdg, ii);
        if (lane == 0) coefs[ii] = aii;
        if (i < ii) ti -= col[ii] * aii;
      }
      const unsigned long long okm = __ballot(ok || lane >= P);
      if (lane == 0) {
        double ldet = 0;
#pragma unroll
        for (int qq = 0; qq < P; qq++) ldet += ldv[qq];
        coefs[P] = ldet;
        coefs[P + 1] = (okm == ~0ull) ? 1.0 : 0.0;
      }
    }
    PIPE_BAR();   // 10
    // ---- slot 12: explicit residual norm ||D - a.ST||^2 (spec_fit.py:249) ------------
    if (!(PIPE_DBG_SKIP & 32)) {
      double av[P];
#pragma unroll
      for (int i = 0; i < P; i++) av[i] = coefs[i];
      double rr = 0;
      for (int k = tid; k < npix; k += PIPE_NC) {
        const double te = tcache[k], dk = tcache[npix + k];
        const double *prow = S.polysT + (int64_t)k * P;
        double mdl = 0;
#pragma unroll
        for (int i = 0; i < P; i++) mdl = fma(av[i], prow[i], mdl);
        const double r = dk - mdl * te;
        rr = fma(r, r, rr);
      }
      rr = wave_sum(rr);
      if (lane == 0) red8[w] = rr;
    }
    PIPE_BAR();   // 11
    // ---- slot 13: the item's outputs ---------------------------------------------------
    if (tid == 0) {
      const int64_t o = (int64_t)arm * J + j;
      if (!usable) {   // unusable template: arm skipped
        armout[o] = __builtin_nan("");
        armchi[o] = 0.0;
        armst[o] = 0;
      } else {
        double rr = red8[0];
        for (int qq = 1; qq < PIPE_NCW; qq++) rr += red8[qq];
        const double lz = S.work[npix + 2ll * S.S * npix + 2 * s];
        double chi = 2.0 * coefs[P] + 2.0 * lz + rr;
        int st = st_extra;
        const double xa = S.lam[0] * f, xb = S.lam[npix - 1] * f;
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
        armchi[o] = chi;
        armst[o] = st;
        armout[o] = outside;
      }
    }
    PIPE_BAR();   // 12
  }
}

// largest template grid the persistent kernel holds in LDS for this npoly
int objective_pipe_max_ntp(int npoly) {
  hipFuncAttributes at;
  const void *fn = nullptr;
#define RVS_CASE(PP)                                      \
  case PP:                                                \
    fn = (const void *)objective_pipe_kernel<PP, false>;  \
    break;
  switch (npoly) {
    RVS_ALL_CASES
    default:
      return 0;
  }
#undef RVS_CASE
  if (hipFuncGetAttributes(&at, fn) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  const int64_t room = 160 * 1024 - (int64_t)at.sharedSizeBytes;
  int n = (int)(room / (3 * (int64_t)sizeof(double)));
  return n > 8192 ? 8192 : (n < 0 ? 0 : n);
}

static int pipe_cu_count() {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) !=
            hipSuccess || v < 1) {
      (void)hipGetLastError();
      v = 256;
    }
    ncu = v;
  }
  return ncu;
}

int objective_pipe_launch(const ObjArms &A, const ObjTempl *tt, int npoly,
                          const double *locrec, const double *vsini,
                          const int32_t *job_spec, int J, const double *vel,
                          size_t nmax, double *armchi, int32_t *armst,
                          double *armout, hipStream_t st) {
  // covered: every arm's chi^2 terms fit the buffer the spline factors leave
  // (2 npix <= ntp: the template is sampled at least twice per pixel), grids of
  // up to 4 dimensions (the cell records of objective_locate_kernel)
  for (int i = 0; i < A.n; i++) {
    if (2 * A.a[i].pt.npix > A.a[i].ntp) return RVS_E_ARG;
    if (!tt && (!locrec || (1 << A.a[i].ndim) > OBJ_LOC_NV)) return RVS_E_ARG;
  }
  if ((int)nmax > objective_pipe_max_ntp(npoly)) return RVS_E_ARG;
  ObjTempl TT = {};
  if (tt) TT = *tt;
  const int64_t total = (int64_t)J * A.n;
  if (total > 0x7fffffff) return RVS_E_ARG;
  const int nblk = (int)(total < pipe_cu_count() ? total : pipe_cu_count());
  const size_t shm = 3 * nmax * sizeof(double);
#define RVS_LAUNCH_PIPE(PP, FT)                                                   \
  {                                                                               \
    static bool attr_set = false;                                                 \
    if (!attr_set) {                                                              \
      (void)hipFuncSetAttribute((const void *)objective_pipe_kernel<PP, FT>,      \
                                hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                160 * 1024 - 1024);                               \
      (void)hipGetLastError();                                                    \
      attr_set = true;                                                            \
    }                                                                             \
    hipLaunchKernelGGL((objective_pipe_kernel<PP, FT>), dim3(nblk), dim3(PIPE_NT), \
                       shm, st, A, TT, locrec, vsini, job_spec, J, (int)nmax,     \
                       vel, 0.6, armchi, armst, armout);                          \
  }
#define RVS_CASE(PP)                                                    \
  case PP:                                                              \
    if (tt) RVS_LAUNCH_PIPE(PP, true) else RVS_LAUNCH_PIPE(PP, false)   \
    break;
  switch (npoly) {
    RVS_ALL_CASES
    default:
      return RVS_E_ARG;
  }
#undef RVS_CASE
#undef RVS_LAUNCH_PIPE
  if (hipGetLastError() != hipSuccess) return RVS_E_LAUNCH;
  return 0;
}
