// objective.hip -- the optimiser's objective, get_chisq (spec_fit.py:797-989) at
// ONE (template parameters, vsini, velocity) point per job, as a single kernel.
//
// vel_fit.process evaluates chisq_func ~850 times per spectrum, each time with
// a NEW template (vel_fit.py:205-254).  Built from the stand-alone kernels this
// is a chain of ~12 small dependent launches per evaluation (gather, FIR, spline
// solve, records to HBM and back, chi^2), and the optimiser's rounds end up
// bound by that chain's latency.  Here one 512-thread block owns a (job, arm)
// and keeps the whole template in LDS (3 x ntp doubles = 155 KB for the DESI z
// arm, which is why this only exists on a 160-KB-LDS part):
//   A  <- polylinear gather of the 2^ndim float32 rows + exp     (A3, A5)
//   B  <- rotational-broadening FIR of A, taps built in C        (A6)
//   dp <- natural spline of y: every thread solves a chunk of rows in registers
//         (factors from rvs_spline_factors' chunk-ordered copy), chunks joined by
//         their transfer coefficients                            (A7 construct)
//   chi^2 <- threads = pixels: spline value from (y, z) on the fly, normal
//         equations per lane, wave butterflies + fixed-order fold, Cholesky,
//         explicit residual norm                                  (A7 eval, A10, A11)
// At two waves per SIMD every vector instruction costs the block the same ~2 ns and
// every vector-memory request of a burst ~45 ns whatever it carries: the
// kernel is written to the COUNT of both -- no per-element range tests (zero pads in
// LDS), scalars of the job computed by one lane, factors as 32-byte records.
// No spline record ever goes to HBM.  Each phase repeats the arithmetic of the
// stand-alone kernel it replaces (template.hip, chisq.hip), so the values agree
// to rounding (tests/test_gpu_parity.py::test_objective_fused).
#include "objective_dev.h"
#include <stdlib.h>
#include <type_traits>

#ifndef OBJ_NT
#define OBJ_NT 512
#endif
#define OBJ_NW (OBJ_NT / 64)
static_assert(OBJ_NT != 512 || (OBJ_NT == RVS_OBJ_NT), "chunk geometry");
#ifndef OBJ_SORT_MIN_JOBS
#define OBJ_SORT_MIN_JOBS 512   // launches below this keep the caller's job order
#endif
#ifdef OBJ_EXP_ONEROW   // (measurement only: every vertex load hits row 0's lines)
#define OBJ_VTX(u) 0
#else
#define OBJ_VTX(u) min(u, nv - 1)
#endif
// rows of a thread's Thomas chunk (at least the 12 the transfer coefficients need)
#define OBJ_CHMAX ((8192 + OBJ_NT - 1) / OBJ_NT > 12 ? (8192 + OBJ_NT - 1) / OBJ_NT : 12)
#ifndef OBJ_FIR_PAD
// doubles of zeros kept on both sides of the template buffer: the register-window FIR
// reads its inputs without range tests
#define OBJ_FIR_PAD OBJ_FIR_KMAX
#endif
#ifndef OBJ_VROW_LANES
#define OBJ_VROW_LANES 1   // row bases: one LDS read + v_readlane (0: a read per vertex)
#endif
#ifndef OBJ_FIR_CHUNKS
// the register-window FIR works in the spline's chunks and hands its outputs over in
// registers (0: its own chunks, a barrier and LDS reads in between)
#define OBJ_FIR_CHUNKS 1
#endif
#ifndef OBJ_FT_IN_FIR
#define OBJ_FT_IN_FIR 1   // spline factor requests between the FIR's outputs (0: ahead)
#endif
#ifndef OBJ_FIR_REG
#define OBJ_FIR_REG 1
#endif

#ifdef RVS_OBJ_TIMING
// debug build only (tools/perf/obj_phases.sh): clock budget of the phases
__device__ unsigned long long obj_dbg[24];
#define OBJ_T(i)                                                         \
  do {                                                                   \
    __syncthreads();                                                     \
    if (threadIdx.x == 0) {                                              \
      const unsigned long long t_ = wall_clock64();                      \
      atomicAdd(&obj_dbg[i], t_ - t_prev);                               \
      t_prev = t_;                                                       \
    }                                                                    \
  } while (0)
extern "C" int rvs_dbg_read(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(obj_dbg), sizeof(obj_dbg)) ==
                 hipSuccess
             ? 0
             : -1;
}
// (no barrier: thread 0's own clock inside a phase that one wave runs)
#define OBJ_TW(i)                                                        \
  do {                                                                   \
    if (threadIdx.x == 0) {                                              \
      const unsigned long long t_ = wall_clock64();                      \
      atomicAdd(&obj_dbg[i], t_ - t_prev);                               \
      t_prev = t_;                                                       \
    }                                                                    \
  } while (0)
#else
#define OBJ_T(i)
#define OBJ_TW(i)
#endif

// (Two waves: the cell search itself needs 20 threads, but a point outside the grid --
// an optimiser's simplex is there every few steps -- scans all ngrid nodes for its
// nearest neighbour, and the launch lasts as long as its slowest block: with one wave
// per block a 17 600-node library cost every objective launch ~100 us.  Four waves, of
// which two had nothing to do, halved the blocks a CU holds: `--process 10000` 3035
// against 3120 spectra/s with two, the 17 600-node library equal.)
#ifndef OBJ_LOC_NT
#define OBJ_LOC_NT 128   // (two waves: the cell search, and the job's scalars and taps)
#endif
__global__ void __launch_bounds__(OBJ_LOC_NT)
    objective_locate_kernel(ObjArms A, const double *__restrict__ params, int J,
                            const int32_t *__restrict__ live,
                            const double *__restrict__ vel,
                            const double *__restrict__ vsini, double eps_ld,
                            double *__restrict__ loc) {
  __shared__ PolyLoc PL;
  const rvs_objective_arm &T = A.a[blockIdx.y];
  const int j = blockIdx.x, tid = threadIdx.x;
  if (live && j >= live[0]) return;   // (see objective_kernel)
  const GridDesc G = obj_grid_desc(T);
  poly_locate<OBJ_LOC_NT>(PL, G, params + (int64_t)j * T.ndim, T.idgrid, T.uvecs,
                          T.vecs_s, T.ngrid);
  double *r = loc + ((int64_t)blockIdx.y * J + j) * OBJ_LOC_REC;
  if (tid < OBJ_LOC_NV) {
    r[tid] = PL.w[tid];
    reinterpret_cast<int64_t *>(r)[OBJ_LOC_NV + tid] = PL.id[tid];
  } else if (tid == OBJ_LOC_NV) {
    r[2 * OBJ_LOC_NV] = PL.dist;
    int32_t *mi = reinterpret_cast<int32_t *>(r + 2 * OBJ_LOC_NV + 1);
    mi[0] = PL.mode;
    mi[1] = PL.nearest;
  } else if (tid == 64 + 32) {   // (the other wave, beside the record's stores)
    obj_job_scalars(T.pt, vel[j], r + 2 * OBJ_LOC_NV + 2);
  }
  // The taps of a narrow rotational kernel (half width <= OBJ_FIR_KMAX: what the
  // optimiser's jobs have on the DESI lattice up to ~150 km/s): in the objective
  // block their construction -- an asin and a sqrt per primitive, ~400 dependent
  // instructions on a dozen lanes behind three barriers -- sat between the first
  // vertex rows' request and the gather, longer than those rows take to arrive.
  // Same functions, same order of the sum (one wave's butterfly; the block's other
  // waves add zeros): the same taps.
  __shared__ double tpk[2][OBJ_FIR_KMAX + 3];
  double R = 0;
  bool refused = false;
  const int kmax = vsini ? obj_rot_kmax(T, vsini[j], R, refused) : 0;
  if (tid == 64 + 33) {
    int32_t *ri = reinterpret_cast<int32_t *>(r + OBJ_LOC_ROT);
    ri[0] = kmax;
    ri[1] = refused ? 1 : 0;
  }
  if (kmax >= 1 && kmax <= OBJ_FIR_KMAX) {   // (the block's: no divergence)
    const int l = tid - 64;   // (lanes of the second wave; 32 and 33 have other work)
    if (l >= 0 && l <= kmax + 2) {
      const double x = fmin(fmax((l - 1) / R, -1.0), 1.0);
      double k0, k1;
      rot_prim(x, eps_ld, k0, k1);
      tpk[0][l] = k0;
      tpk[1][l] = k1;
    }
    __syncthreads();
    if (l >= 0 && l < 64) {
      const double ww = (l <= kmax) ? obj_rot_tap_raw(l, R, tpk[0], tpk[1]) : 0.0;
      const double psum = wave_sum((l == 0) ? ww : 2 * ww);
      const double inv = 1.0 / psum;
      if (l <= kmax) r[OBJ_LOC_TAPS + l] = ww * inv;
    }
  }
}

// Jobs of a launch in the order of their grid cell: a counting sort of the lowest
// vertex row of arm 0's cell record (the nearest node for a point outside the
// grid) by one 1024-thread block -- histogram in LDS, block scan, scatter.  The
// order inside a bin is whatever the atomics give; it only decides WHICH block
// evaluates a job.
#define OBJ_ORD_NT 1024
#define OBJ_ORD_NB 8192
__global__ void __launch_bounds__(OBJ_ORD_NT)
    objective_order_kernel(const double *__restrict__ loc, int J,
                           const int32_t *__restrict__ live, int shift, int nb,
                           int32_t *__restrict__ perm) {
  const int Jbound = J;
  if (live) J = min(J, live[0]);   // the jobs that count are the first live[0]
  __shared__ int hist[OBJ_ORD_NB];
  __shared__ int wsum[OBJ_ORD_NT / 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < nb; i += OBJ_ORD_NT) hist[i] = 0;
  __syncthreads();
  auto key_of = [&](int j) {
    const double *r = loc + (int64_t)j * OBJ_LOC_REC;
    const int32_t *mi = reinterpret_cast<const int32_t *>(r + 2 * OBJ_LOC_NV + 1);
    const int64_t id = (mi[0] == 0) ? reinterpret_cast<const int64_t *>(r)[OBJ_LOC_NV]
                                    : (int64_t)mi[1];
    const int64_t k = (id < 0 ? 0 : id) >> shift;
    return (int)(k < nb ? k : nb - 1);
  };
  // (a thread's keys stay in registers for the scatter pass -- up to 8 x 1024 jobs;
  // read again behind that: two dependent loads per job)
  constexpr int KPT = 8;
  int keys[KPT];
#pragma unroll
  for (int i = 0; i < KPT; i++) {
    const int j = tid + i * OBJ_ORD_NT;
    keys[i] = (j < J) ? key_of(j) : 0;
    if (j < J) atomicAdd(&hist[keys[i]], 1);
  }
  for (int j = tid + KPT * OBJ_ORD_NT; j < J; j += OBJ_ORD_NT) atomicAdd(&hist[key_of(j)], 1);
  __syncthreads();
  // exclusive scan: every thread owns nb / 1024 consecutive bins
  const int per = (nb + OBJ_ORD_NT - 1) / OBJ_ORD_NT;
  const int b0 = min(nb, tid * per), b1 = min(nb, b0 + per);
  int loc_sum = 0;
  for (int i = b0; i < b1; i++) loc_sum += hist[i];
  int inc = loc_sum;   // inclusive scan over the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int base = 0;
  for (int q = 0; q < w; q++) base += wsum[q];
  int run = base + inc - loc_sum;
  for (int i = b0; i < b1; i++) {
    const int c = hist[i];
    hist[i] = run;
    run += c;
  }
  __syncthreads();
  // Position `pos` of the sorted list goes to block x = 8 i + f, the i-th block of XCD f
  // (blocks are dealt to the 8 XCDs round robin; XCD f works through the f-th
  // contiguous eighth of the list, objective_kernel): the table is written by BLOCK
  // NUMBER, so that a block finds its job with one load of an address it knows when it
  // starts -- not live[0], then the list's entry at a position that depends on it.
  // Blocks behind the live jobs read -1.
  const int base8 = J >> 3, rem8 = J & 7;
  for (int jj = 0; jj * OBJ_ORD_NT < J; jj++) {
    const int j = tid + jj * OBJ_ORD_NT;
    if (j >= J) break;
    int key = 0;
    bool have = false;
#pragma unroll
    for (int i = 0; i < KPT; i++)
      if (i == jj) key = keys[i], have = true;
    if (!have) key = key_of(j);
    const int pos = atomicAdd(&hist[key], 1);
    int f, i;
    if (pos < rem8 * (base8 + 1)) {
      f = pos / (base8 + 1);
      i = pos % (base8 + 1);
    } else {
      const int q = pos - rem8 * (base8 + 1);
      f = rem8 + q / base8;
      i = q % base8;
    }
    perm[8 * i + f] = j;
  }
  for (int x = J + tid; x < Jbound; x += OBJ_ORD_NT) perm[x] = -1;
}

// INBLK: grids of more than 4 dimensions (no cell record: the search runs in
// the block; its dynamically indexed descriptor costs every instantiation that
// contains it 168 B of scratch per lane, so it is a variant of its own)
template <int P, bool FROMT, bool INBLK = false>
__global__ void __launch_bounds__(OBJ_NT)
    objective_kernel(ObjArms A, ObjTempl TT, const double *__restrict__ locrec,
                     const int32_t *__restrict__ perm,
                     const int32_t *__restrict__ live,
                     const double *__restrict__ params,
                     const double *__restrict__ vsini,
                     const int32_t *__restrict__ job_spec, int J,
                     const double *__restrict__ vel, double eps_ld,
                     double *__restrict__ armchi, int32_t *__restrict__ armst,
                     double *__restrict__ armout) {
  constexpr int NT = P * (P + 1) / 2;
  constexpr int NV = NT + P;
  extern __shared__ double lds[];
  __shared__ PolyLoc PL;
  // wave totals of the NV sums.  Up to P = 10 a static array; beyond (P = 15: 8.7 KB,
  // which took the DESI z arm's 6449 knots out of the kernel's reach in round 3:
  // `--npoly 15 --process` ran the kernel chain, 452 spectra/s) they live in the
  // template buffer, which is dead by the time they are formed -- such an
  // instantiation needs the sigma-scaled model to fit the factor buffer (2 npix <=
  // ntp, checked by the launcher)
  constexpr bool RED_DYN = (P > 10) || (OBJ_NT > 512);
  __shared__ double red_static[RED_DYN ? 1 : OBJ_NW * (NV + 1)];
  __shared__ double edge_s[2][OBJ_NW][6];   // chunk coefficients across wave boundaries
  __shared__ double coefs[P + 2];
  __shared__ double Lm[P + 1][P + 1];   // (row P: y of L y = v)
  __shared__ double ldv[P];
  __shared__ double dgv[P];    // the diagonal of L on its way to the wave that takes its log
  __shared__ int dg_flag;     // 1: dgv[] is there
  __shared__ double red8[2 * OBJ_NW];
  __shared__ double jobsc[3];   // the job's Doppler scalars (obj_job_scalars)
  __shared__ int rot_s[2];      // {kmax, refused} of its rotational kernel (the record's)
  const rvs_objective_arm &T = A.a[blockIdx.y];
  const int tid = threadIdx.x;
  if (tid == 0) dg_flag = 0;   // (many barriers ahead of its use)
  // Job of this block.  With `perm` (the jobs of the launch in the order of their
  // grid cell, objective_order_kernel) block x takes position
  //   p = f (J / 8) + min(f, J % 8) + (x >> 3),   f = x & 7:
  // blocks are dealt to the 8 XCDs round robin, so the blocks of one f share an
  // XCD and work through one contiguous eighth of the sorted list -- the vertex
  // rows of a cell (shared by all jobs in it, and half of them by the next cell)
  // are fetched into that XCD's L2 once and hit there afterwards.  Which block
  // computes a job does not change its value.
  int j = blockIdx.x;
  // `live` (device): only the first live[0] of the launch's J jobs are evaluations
  // somebody waits for.  The lock-step optimiser sizes its launches by the count of
  // running simplices at its last look (every few rounds, the only host
  // synchronisation) and by all of them for the second point of a round, which 60-70 %
  // need: a quarter of the blocks of a Nelder-Mead run computed values nobody read
  // (tools/perf/nm_waste.py: 1.39 launched slots per function value scipy counts).
  // Such a block ends here; the array strides stay those of J.
  if (perm) {
    // (the order kernel's table by block number: -1 behind the live jobs; one value
    // for the block, kept in a scalar register)
    j = __builtin_amdgcn_readfirstlane(perm[j]);
    if (j < 0) return;
  } else {
    const int Jl = live ? min(J, live[0]) : J;
    if (j >= Jl) return;
  }
  const int lane = tid & 63, w = tid >> 6;
  const int N = T.ntp, m = N - 2;
  // [pad][bufA: N][pad][bufB: N][bufC: N]
  double *bufA = lds + OBJ_FIR_PAD, *bufB = bufA + N + OBJ_FIR_PAD, *bufC = bufB + N;
  if (OBJ_FIR_PAD && tid < 2 * OBJ_FIR_PAD)   // (barriers follow before the FIR reads)
    (tid < OBJ_FIR_PAD ? lds : bufA + N - OBJ_FIR_PAD)[tid] = 0.0;
  double (*red)[NV + 1] =
      reinterpret_cast<double (*)[NV + 1]>(RED_DYN ? bufA : red_static);
  const int nd = T.ndim, nv = 1 << nd;
#ifdef RVS_OBJ_TIMING
  unsigned long long t_prev = wall_clock64();
#endif
  // ---- A3/A5: polylinear template into bufA -------------------------------
  int mode = 0;
  double outside_in = 0.0;
  if (!FROMT) {
    if (locrec) {   // the record objective_locate_kernel left for this (job, arm)
      const double *r = locrec + ((int64_t)blockIdx.y * J + j) * OBJ_LOC_REC;
      if (tid < OBJ_LOC_NV) {
        PL.w[tid] = r[tid];
        PL.id[tid] = reinterpret_cast<const int64_t *>(r)[OBJ_LOC_NV + tid];
      } else if (tid == 64) {
        PL.dist = r[2 * OBJ_LOC_NV];
        const int32_t *mi =
            reinterpret_cast<const int32_t *>(r + 2 * OBJ_LOC_NV + 1);
        PL.mode = mi[0];
        PL.nearest = mi[1];
      } else if (tid >= 128 && tid < 131) {
        jobsc[tid - 128] = r[2 * OBJ_LOC_NV + 2 + tid - 128];
      } else if (tid == 192) {
        const int32_t *ri = reinterpret_cast<const int32_t *>(r + OBJ_LOC_ROT);
        rot_s[0] = ri[0];
        rot_s[1] = ri[1];
      }
      __syncthreads();
    } else if (INBLK) {
      const GridDesc G = obj_grid_desc(T);
      poly_locate<OBJ_NT>(PL, G, params + (int64_t)j * nd, T.idgrid, T.uvecs,
                          T.vecs_s, T.ngrid);
    }
    mode = PL.mode;
  } else {
    // outside != 0 (or not finite): the MAX_VAL guard below scans the row
    outside_in = TT.outside[blockIdx.y][j];
    mode = (outside_in == 0.0) ? 0 : 1;
  }
  OBJ_T(0);
  double mx = 0;
  bool anynan = false;
  // The first pixel group's vertex rows are requested NOW and the rotational
  // kernel is built while they are in flight (it depends on the job's vsini
  // only: ~1.5 us of asin / sqrt arithmetic and two barriers that used to sit
  // behind the gather, between its last round trip and the FIR).
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  const int N4 = N & ~3;
  const bool vec_gather = !FROMT && mode == 0 && nv <= 16;
  f4u rn[16];
  // the vertex rows' base addresses are the block's, not the lane's: formed once, in
  // scalar registers (inside the loop each of the 16 loads of a trip re-read its row
  // number from LDS and multiplied it out in 64-bit vector arithmetic: ~9 of the
  // ~25 vector instructions per load)
  const float *vrow[16];
  if (vec_gather) {
#if OBJ_VROW_LANES
    // (ONE LDS read: lane u of every wave takes vertex u's row number and forms the
    // row's element offset; sixteen pairs of v_readlane with a constant lane hand them
    // to the scalar side.  One read per vertex, each followed by its wait and two
    // v_readfirstlane, was sixteen LDS round trips in a row ahead of the first request.)
    const int64_t myrow = PL.id[OBJ_VTX(tid & 15)] * (int64_t)N;
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)myrow, u);
      const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(myrow >> 32), u);
      vrow[u] = T.dats + (int64_t)(((uint64_t)hi << 32) | lo);
    }
#else
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int64_t id = PL.id[OBJ_VTX(u)];
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)id);
      const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(id >> 32));
      vrow[u] = T.dats + (int64_t)(((uint64_t)hi << 32) | lo) * N;
    }
#endif
  }
  if (vec_gather && 4 * tid < N4) {
#pragma unroll
    for (int u = 0; u < 16; u++)
      rn[u] = *reinterpret_cast<const f4u *>(vrow[u] + 4 * tid);
  }
  if ((FROMT || !locrec) && tid == 64) obj_job_scalars(T.pt, vel[j], jobsc);
  OBJ_T(15);   // (debug) row bases, first group requested
  if (FROMT) {
    // The block's template row -- 50 KB of float64 the evaluator's kernel left in HBM
    // -- by LDS-DMA (global_load_lds: no registers, nothing to wait for until the
    // next barrier), every wave's share requested here in one go, in flight under
    // the rotational kernel's construction.  (As a loop of load -> LDS store behind
    // that construction the compiler kept a few loads in flight at a time and the
    // row took longer than the 16-row gather it replaces: 46 against 34 us per block
    // in the optimiser's rounds on an MLP library.)  Rows start on 8-byte
    // boundaries only, so the pieces are dwords: wave-uniform LDS base + lane x 4.
    const float *rowf =
        reinterpret_cast<const float *>(TT.templ[blockIdx.y] + (int64_t)j * N);
    float *dstf = reinterpret_cast<float *>(bufA);
    const int ndw = 2 * N;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int c0 = wv * 64; c0 < ndw; c0 += OBJ_NT)
      if (c0 + (tid & 63) < ndw)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(rowf + c0 + (tid & 63)),
            (__attribute__((address_space(3))) void *)(dstf + c0), 4, 0, 0);
  }
  int st_extra = 0;
  bool copy = true;
  int kmax = 0;
  if (vsini) {
    double R = 0;
    bool refused;
    const bool rec = !FROMT && !INBLK && locrec;
    if (rec) {   // (two divisions per thread less: the cell-search kernel's values)
      kmax = __builtin_amdgcn_readfirstlane(rot_s[0]);
      refused = rot_s[1] != 0;
      if (kmax > OBJ_FIR_KMAX) kmax = obj_rot_kmax(T, vsini[j], R, refused);   // (R)
    } else {
      kmax = obj_rot_kmax(T, vsini[j], R, refused);
    }
    copy = kmax == 0;
    if (refused) st_extra = RVS_ST_NONFINITE;
    if (!copy && rec && kmax <= OBJ_FIR_KMAX) {
      // (the cell-search kernel built them: objective_locate_kernel; the FIR reads
      // them behind the gather's barriers)
      if (tid <= kmax)
        bufC[tid] = (locrec + ((int64_t)blockIdx.y * J + j) * OBJ_LOC_REC)[OBJ_LOC_TAPS + tid];
    } else if (!copy) {
      // The kernel's primitives (an asin and a sqrt each, ~150 dependent fp64
      // instructions) at the clipped points x_j = clip(j / R), j = -1 .. kmax+1,
      // ONE per thread: tap k needs them at j = k-1, k, k+1, and evaluated
      // inside rot_segment every tap thread walked four of them in a row while
      // the other 450 threads of the block waited.  Same arguments, same
      // function: the same values.  (bufB is free until the FIR writes it.)
      double *pk0 = bufB, *pk1 = bufB + (kmax + 3);
      for (int jj = tid; jj <= kmax + 2; jj += OBJ_NT) {
        const double x = fmin(fmax((jj - 1) / R, -1.0), 1.0);
        double k0, k1;
        rot_prim(x, eps_ld, k0, k1);
        pk0[jj] = k0;
        pk1[jj] = k1;
      }
      __syncthreads();
      double psum = 0;
      for (int k = tid; k <= kmax; k += OBJ_NT) {
        const double ww = obj_rot_tap_raw(k, R, pk0, pk1);
        bufC[k] = ww;
        psum += (k == 0) ? ww : 2 * ww;
      }
      psum = block_sum<OBJ_NW>(psum, red8);
      __syncthreads();
      const double inv = 1.0 / psum;
      // normalised taps once (the product every output formed per tap)
      for (int k = tid; k <= kmax; k += OBJ_NT) bufC[k] = bufC[k] * inv;
      __syncthreads();
    }
  }
  OBJ_T(16);   // (debug) rotational kernel built
  if (FROMT) {
    // (the row is on its way into bufA: LDS-DMA below, complete at the next barrier)
    if (mode != 0) {   // MAX_VAL guard: the scan needs the values
      __syncthreads();
      for (int k = tid; k < N; k += OBJ_NT) {
        const double val = bufA[k];
        if (!(val == val)) anynan = true;
        mx = fmax(mx, fabs(val));
      }
    }
  } else if (mode == 0) {
    // four CONSECUTIVE pixels per thread: one 16-byte load per vertex row
    // (rows start on 4-byte boundaries only: dword-aligned x4 loads), 1 KiB per
    // wave instruction instead of 256 B
    if (nv <= 16) {
      // up to 4-D grids: the 2^ndim vertex rows of a pixel group are requested
      // together, and the group after it before this one is consumed (one L2
      // round trip per group, hidden behind the conversions, FMAs and exps of
      // the previous group); the sums run in vertex order.  (First group:
      // requested above, ahead of the rotational kernel.  Two groups in flight --
      // 243 VGPRs -- measured: 5.10 against 5.00 s of Nelder-Mead per 10 000
      // spectra, the phase is not waiting for latency.)
      // The vertex weights are the block's: in registers for the whole phase, zero
      // behind the grid's 2^ndim (such a slot reads the last vertex row again and adds
      // an exact zero).  Read from LDS at their use -- one ds_read + wait per vertex
      // and trip, behind a scalar test of u < nv that also kept the sixteen blends
      // of a trip from interleaving -- they were 64 serialised LDS round trips per
      // block.
      double wreg[16];
#pragma unroll
      for (int u = 0; u < 16; u++) wreg[u] = (u < nv) ? PL.w[u] : 0.0;

      // Two register sets take turns (a trip loads into one and blends the other): as
      // one set copied into a second at the head of every trip the compiler moved 64
      // dwords per trip.  The rows' base addresses are scalars and the offset a 32-bit
      // byte count, which is the addressing a global load has (no 64-bit address
      // arithmetic per load).
      auto issue = [&](f4u *dst, int k) {
        const uint32_t boff = (uint32_t)k * 4u;
#pragma unroll
        for (int u = 0; u < 16; u++)
          dst[u] = *reinterpret_cast<const f4u *>(
              reinterpret_cast<const char *>(vrow[u]) + boff);
      };
      auto blend = [&](const f4u *r, int k) {
        double a4[4] = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const double wv = wreg[u];
          a4[0] = fma(wv, (double)r[u].x, a4[0]);
          a4[1] = fma(wv, (double)r[u].y, a4[1]);
          a4[2] = fma(wv, (double)r[u].z, a4[2]);
          a4[3] = fma(wv, (double)r[u].w, a4[3]);
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
          bufA[k + q] = T.exp_flag ? exp(a4[q]) : a4[q];
      };
      for (int k = 4 * tid; k < N4; k += 4 * OBJ_NT) {
        f4u r[16];
#pragma unroll
        for (int u = 0; u < 16; u++) r[u] = rn[u];
        const int kn = k + 4 * OBJ_NT;
        if (kn < N4) issue(rn, kn);
        blend(r, k);
      }
    } else {
      for (int k = 4 * tid; k < N4; k += 4 * OBJ_NT) {
        double a4[4] = {0, 0, 0, 0};
        for (int v0 = 0; v0 < nv; v0 += 16) {
          f4u r[16];
#pragma unroll
          for (int u = 0; u < 16; u++) {
            const int v = min(v0 + u, nv - 1);
            r[u] = *reinterpret_cast<const f4u *>(T.dats + PL.id[v] * N + k);
          }
#pragma unroll
          for (int u = 0; u < 16; u++) {
            if (v0 + u < nv) {
              const double wv = PL.w[v0 + u];
              a4[0] = fma(wv, (double)r[u].x, a4[0]);
              a4[1] = fma(wv, (double)r[u].y, a4[1]);
              a4[2] = fma(wv, (double)r[u].z, a4[2]);
              a4[3] = fma(wv, (double)r[u].w, a4[3]);
            }
          }
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
          bufA[k + q] = T.exp_flag ? exp(a4[q]) : a4[q];
      }
    }
    for (int k = N4 + tid; k < N; k += OBJ_NT) {
      double acc = 0;
      for (int v = 0; v < nv; v++)
        acc = fma(PL.w[v], (double)T.dats[PL.id[v] * N + k], acc);
      bufA[k] = T.exp_flag ? exp(acc) : acc;
    }
  } else {
    const float *row = T.dats + (int64_t)PL.nearest * N;
    for (int k = tid; k < N; k += OBJ_NT) {
      const double val = T.exp_flag ? (double)np_expf(row[k])
                                    : (double)row[k];
      bufA[k] = val;
      if (!(val == val)) anynan = true;
      mx = fmax(mx, fabs(val));
    }
  }
  OBJ_T(17);   // (debug) gather loop
  double outside = 0.0;
  if (mode != 0) {  // MAX_VAL guard of getCurTempl (spec_fit.py:392-397)
    mx = wave_max(mx);
    const double nanf = wave_sum(anynan ? 1.0 : 0.0);
    if (lane == 0) {
      red8[w] = mx;
      red8[OBJ_NW + w] = nanf;
    }
    __syncthreads();
    double mm = 0, nn = 0;
    for (int i = 0; i < OBJ_NW; i++) {
      mm = fmax(mm, red8[i]);
      nn += red8[OBJ_NW + i];
    }
    outside = FROMT ? outside_in : PL.dist;
    if (outside > 0 && (mm > 1e100 || nn > 0 || isinf(mm)))
      outside = __builtin_nan("");
  }
  __syncthreads();
  if (!(fabs(outside) <= 1.79e308)) {  // unusable template: arm skipped
    if (tid == 0) {
      armout[(int64_t)blockIdx.y * J + j] = __builtin_nan("");
      armchi[(int64_t)blockIdx.y * J + j] = 0.0;
      armst[(int64_t)blockIdx.y * J + j] = 0;
    }
    return;
  }
  OBJ_T(1);
  const rvs_point_arm &S = T.pt;
  const int npix = S.npix;
  const int s = job_spec ? job_spec[j] : j;
  const ObjArmGrid AG = obj_arm_grid(S, s);   // the spectrum's wavelength grid
  // {1/e, s/e} of the spectrum's pixels (rvs_chisq_prepare's second table: e with the
  // systematic floor of `espec_sys` in quadrature, 0 / 0 on the padding of a short
  // grid): one 16-byte load where every evaluation used to take a square root, a
  // division and a product per pixel -- the same operations, done once per batch
  const double2 *sig = reinterpret_cast<const double2 *>(
                           AG.wbase + 2ll * S.S * npix + 2ll * S.S) + (int64_t)s * npix;
  const double x0 = S.knots[0], xlast = S.knots[N - 1];
  const double *g = T.factors, *e = T.factors + N, *cc = T.factors + 2 * N,
               *hh = T.factors + 3 * N, *ih = T.factors + 4 * N;
  auto knot_pair = [&](int pos) {   // knots pos, pos + 1 (rows start on 8 bytes)
    typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
    const d2u v = *reinterpret_cast<const d2u *>(S.knots + pos);
    return make_double2(v.x, v.y);
  };
  // The phases below are separated by barriers and the block is the only one on its
  // CU (two waves per SIMD): a load issued where its value is needed is a fully
  // exposed L2 round trip.  The spline factors of this thread's CHUNK of rows (row
  // a0 + q at [q][tid] of the chunk-transposed arrays: coalesced) are requested HERE,
  // ahead of the FIR (which needs few registers), the backward multipliers under the
  // forward sweep, the model pass's pixel terms under the backward sweep.
  static_assert(OBJ_NT == RVS_OBJ_NT && OBJ_CHMAX == RVS_OBJ_CHMAX, "chunk geometry");
  const int CH = __builtin_amdgcn_readfirstlane(rvs_obj_chunk_len(m));
  // ({1/h_u, 1/h_{u+1}, g_u, e_u} is one 32-byte record per row, the backward
  // multipliers come in pairs of rows: 33 requests of 16 bytes per thread where five
  // arrays of doubles took 65 -- at this occupancy a request costs what it costs
  // whatever its width, ~45 ns of the block each)
  const double *FT = T.factors + 5 * (int64_t)N;
  typedef double d2v __attribute__((ext_vector_type(2)));
  auto ft16 = [&](int64_t base, uint32_t piece) {   // 16-byte piece of a [..][tid] table
    return *reinterpret_cast<const d2v *>(
        reinterpret_cast<const char *>(FT + base) + piece * 16u);
  };
  double tf0[OBJ_CHMAX], tf1[OBJ_CHMAX], tfg[OBJ_CHMAX], tfe[OBJ_CHMAX];
  auto ft_fetch = [&](int q) {   // (q < OBJ_CHMAX: a compile-time slot)
    if (q < CH) {
      const d2v a = ft16(0, (uint32_t)(q * OBJ_NT + tid) * 2u);
      const d2v b = ft16(0, (uint32_t)(q * OBJ_NT + tid) * 2u + 1u);
      tf0[q] = a.x;
      tf1[q] = a.y;
      tfg[q] = b.x;
      tfe[q] = b.y;
    }
  };
  // With the register-window FIR ahead the requests go out BETWEEN its outputs, a
  // row's four after each (fir_small below): issued in one go, the eight waves' 416
  // loads fill the CU's address pipe and every wave sits in its load issue for the
  // 1.4 us the pipe needs before its FIR starts.
  const bool ft_in_fir = OBJ_FT_IN_FIR && vsini && OBJ_FIR_REG && !copy && kmax <= 4;
  if (!ft_in_fir) {
#pragma unroll
    for (int q = 0; q < OBJ_CHMAX; q++) ft_fetch(q);
  }
  // ---- A6: rotational broadening bufA -> bufB (taps in bufC, built above) ----
  double *y = bufA, *dp = bufB;
  bool fir_reg = false, fir_chunks = false;
  double fy[OBJ_CHMAX + 2];   // (the FIR's outputs of this thread's spline chunk)
  if (vsini) {
    // Narrow kernels (kmax <= 8: v sin i up to ~150 km/s on the DESI template lattice)
    // keep the taps and the thread's input window [c0 - KM, c1 + KM) in registers: every
    // input is read from LDS once (the loop below reads it once per group of four
    // outputs, and a tap once per four FMAs).  An output is still the fma chain over
    // ascending offsets; the offsets beyond kmax carry exact zero taps at both ends of
    // the chain, which leave the partial sum as it is -- the values are those of the
    // loop below (36.6 -> 35.1 us per block, same checksum).
    fir_reg = OBJ_FIR_REG && !copy && kmax <= OBJ_FIR_KMAX;
    // (the FIR's chunks are the spline's when 512 CH covers the template)
    fir_chunks = OBJ_FIR_CHUNKS && fir_reg && OBJ_NT * CH >= N;
    auto fir_small = [&](auto km_c) {
      constexpr int KM = decltype(km_c)::value;
      constexpr int LCM = (8192 + OBJ_NT - 1) / OBJ_NT;
      // A thread's outputs are the rows of its spline chunk (CH of them from t CH)
      // and the two behind them: the chunk's right-hand sides then come out of THIS
      // thread's registers (fy), and neither the barrier behind the FIR nor CH + 2 LDS
      // reads stand between the last tap and the forward recurrence.  (The two extra
      // outputs are the next thread's first two, computed twice and stored once.)
      const int Lc = fir_chunks ? CH : (N + OBJ_NT - 1) / OBJ_NT;
      constexpr int LX = OBJ_FIR_CHUNKS ? 2 : 0;
      double tp[KM + 1], win[LCM + LX + 2 * KM];
#pragma unroll
      for (int mm = 0; mm <= KM; mm++) tp[mm] = (mm <= kmax) ? bufC[mm] : 0.0;
      // The inputs behind both ends of the template are zeros IN LDS (the pads): a
      // thread's window is LCM + 2 KM reads off one base address with nothing to test
      // (what lies behind the thread's own Lc + 2 KM is read and not used: the
      // template's three buffers are there: N >= 32 > LCM + KM), and its Lc outputs
      // are computed under the block's own `o < Lc` -- per element under the lane's
      // `0 <= q < N` every read took ~15 instructions of compares and exec masks, 580
      // instructions per thread for 117 FMAs (a phase of 4-5 us with 0.3 us of
      // arithmetic).  A zero that is loaded multiplies like the literal: same values.
      static_assert(KM <= OBJ_FIR_PAD, "pad");
      const int c0 = min(N, tid * Lc);
      const double *wb = bufA + c0 - KM;
#pragma unroll
      for (int i = 0; i < LCM + LX + 2 * KM; i++) win[i] = wb[i];   // (N >= 32: the launcher)
#pragma unroll
      for (int o = 0; o < LCM + LX; o++) {
        if (o < Lc + (fir_chunks ? 2 : 0)) {
          double sacc = 0;
#pragma unroll
          for (int mm = -KM; mm <= KM; mm++)
            sacc = fma(win[o + mm + KM], tp[mm < 0 ? -mm : mm], sacc);
          if (o < Lc && c0 + o < N) bufB[c0 + o] = sacc;
          if (OBJ_FIR_CHUNKS) fy[o] = sacc;
        }
        if constexpr (KM == 4)   // (kmax <= 4: the narrow window leaves the registers)
          if (o < LCM && ft_in_fir) ft_fetch(o);
      }
    };
    if (fir_reg) {
      if (kmax <= 4)
        fir_small(std::integral_constant<int, 4>{});
      else if (kmax <= 8)
        fir_small(std::integral_constant<int, 8>{});
      else
        fir_small(std::integral_constant<int, OBJ_FIR_KMAX>{});
      y = bufB;
      dp = bufA;
      if (!fir_chunks) __syncthreads();
    }
    if (!copy && !fir_reg) {
      // Four consecutive outputs per thread and trip: at tap offset mm the four
      // inputs are a sliding window -- one new LDS read per offset, the tap read
      // once for four FMAs -- where one output per trip read input and tap for
      // every FMA (the phase was LDS bound: 12 % of the block).  Each output
      // still sums over q ascending with separately rounded tap products, so
      // the values are the ones of the plain loop (an input outside [0, N) and a
      // tap offset behind kmax count as exact zeros: fma(0, t, acc) = acc).  Thread t owns
      // the outputs [t L, (t + 1) L).
      {
        const int Lc = (N + OBJ_NT - 1) / OBJ_NT;
        const int c0 = tid * Lc, c1 = min(N, c0 + Lc);
        auto in = [&](int q) { return (q >= 0 && q < N) ? bufA[q] : 0.0; };
        auto tp = [&](int mm) { return bufC[mm < 0 ? -mm : mm]; };
        // W consecutive outputs per trip (8 while that many are left, then 4): per
        // tap offset one new input and one tap are read for W FMAs -- at W = 8 half
        // the LDS reads per FMA of W = 4 (wide kernels: v sin i beyond ~180 km/s on
        // the DESI lattice; the phase is LDS bound there)
        auto fir_trip = [&](auto w_c, int i0) {
          constexpr int W = decltype(w_c)::value;
          double sacc[W], a[W];
#pragma unroll
          for (int o = 0; o < W; o++) sacc[o] = 0;
          int q = i0 - kmax;                 // input of output i0 at offset -kmax
#pragma unroll
          for (int o = 0; o < W; o++) a[o] = in(q + o);
          // W tap offsets per turn of the loop: the window's slots are back in
          // order after W of them (offsets behind kmax carry exact zero taps)
          for (int mm = -kmax; mm <= kmax; mm += W, q += W) {
#pragma unroll
            for (int g = 0; g < W; g++) {
              const double t = (g == 0 || mm + g <= kmax) ? tp(mm + g) : 0.0;
              // offset mm + g: output o takes the window's slot (o + g) mod W
#pragma unroll
              for (int o = 0; o < W; o++) sacc[o] = fma(a[(o + g) % W], t, sacc[o]);
              a[g] = in(q + W + g);   // slot g is free: the next input
            }
          }
#pragma unroll
          for (int o = 0; o < W; o++)
            if (o == 0 || i0 + o < c1) bufB[i0 + o] = sacc[o];
        };
        int i0 = c0;
        for (; i0 + 8 <= c1; i0 += 8) fir_trip(std::integral_constant<int, 8>{}, i0);
        for (; i0 < c1; i0 += 4) fir_trip(std::integral_constant<int, 4>{}, i0);
      }
      y = bufB;
      dp = bufA;
      __syncthreads();
    }
  }
  OBJ_T(2);
  // ---- A7 construct: natural spline of y ---------------------------------------
  // Thread t owns the rows [a0, a1) (CH of them): their right-hand sides
  //   6 ((y_{i+2} - y_{i+1}) / h_{i+1} - (y_{i+1} - y_i) / h_i) g_i
  // from CH + 2 template values read off ONE LDS address (what lies behind a1 is read
  // and not used: the template's buffers are there) and the factors already in its
  // registers, straight into the forward recurrence.  (By strided rows the right-hand
  // sides and multipliers went to LDS and came back in chunk order behind a barrier,
  // the backward multipliers likewise: 130 LDS operations per thread where these are
  // 28, and two barriers.)  Both recurrences are linear in the value that enters the
  // chunk: d_i = d0_i + P_i d_in with d0 the run from zero and P_i the running product
  // of the multipliers, so one pass over the chunk yields (alpha, beta) = (d0, P) at its
  // last row, and
  //   d_in(t) = alpha(t-1) + beta(t-1) (alpha(t-2) + beta(t-2) alpha(t-3))
  // to |beta|^3 <= (0.268^12)^3 = 3e-21 (the multipliers of a (log-)uniform grid tend to
  // 2 - sqrt 3).  Same operations per row as the strided form: the same bits.
  const int a0 = min(m, tid * CH), a1 = min(m, a0 + CH);
  double loc[OBJ_CHMAX], pr[OBJ_CHMAX];
  // value entering a chunk from `dir` = -1 (lower threads) or +1 (upper): the three
  // nearest chunks' coefficients through wave shuffles, across a wave boundary through
  // edge_s[] (static LDS; one array per direction: no barrier between the two uses)
  auto chain3 = [&](double al, double be, int dir) -> double {
    double (*eds)[6] = edge_s[dir < 0 ? 0 : 1];
    double av3[3], bv3[3];
#pragma unroll
    for (int k = 1; k <= 3; k++) {
      av3[k - 1] = (dir < 0) ? __shfl_up(al, k, 64) : __shfl_down(al, k, 64);
      bv3[k - 1] = (dir < 0) ? __shfl_up(be, k, 64) : __shfl_down(be, k, 64);
    }
    const int edge = (dir < 0) ? (63 - lane) : lane;  // 0..2: published lanes
    if (edge < 3) {
      eds[w * 1][2 * edge] = al;
      eds[w * 1][2 * edge + 1] = be;
    }
    __syncthreads();
    const int mine = (dir < 0) ? lane : (63 - lane);  // distance to the boundary
#pragma unroll
    for (int k = 1; k <= 3; k++) {
      if (mine < k) {  // neighbour k lives in the adjacent wave
        const int ww = w + dir;
        const int sl = k - 1 - mine;  // its distance from that wave's boundary
        const bool have = (ww >= 0 && ww < OBJ_NW);
        av3[k - 1] = have ? eds[ww][2 * sl] : 0.0;
        bv3[k - 1] = have ? eds[ww][2 * sl + 1] : 0.0;
      }
    }
    return av3[0] + bv3[0] * (av3[1] + bv3[1] * av3[2]);
  };
  double tfc[OBJ_CHMAX];   // backward multipliers: in flight under the forward sweep
#pragma unroll
  for (int q2 = 0; q2 < OBJ_CHMAX / 2; q2++)
    if (2 * q2 < CH) {
      const d2v c = ft16(4 * OBJ_NT * OBJ_CHMAX, (uint32_t)(q2 * OBJ_NT + tid));
      tfc[2 * q2] = c.x;
      tfc[2 * q2 + 1] = c.y;
    }
  double d_in;
  {
    const double *yb = y + a0;
    double yv[OBJ_CHMAX + 2];
    if (fir_chunks) {
#pragma unroll
      for (int q = 0; q < OBJ_CHMAX + 2; q++) yv[q] = fy[q];
    } else {
#pragma unroll
      for (int q = 0; q < OBJ_CHMAX + 2; q++)
        if (q < CH + 2) yv[q] = yb[q];
    }
    double d = 0, pb = 1;
#pragma unroll
    for (int q = 0; q < OBJ_CHMAX; q++)
      if (q < CH && a0 + q < a1) {
        const double s0 = (yv[q + 1] - yv[q]) * tf0[q],
                     s1 = (yv[q + 2] - yv[q + 1]) * tf1[q];
        const double ei = tfe[q];
        d = 6 * (s1 - s0) * tfg[q] - ei * d;
        pb = -ei * pb;
        loc[q] = d;
        pr[q] = pb;
      }
    OBJ_T(11);   // (debug) right-hand sides + forward recurrence
    d_in = chain3(d, pb, -1);
    OBJ_T(12);   // (debug) chunk hand-over
  }
  // (the job's Doppler scalars, written barriers ago by one lane)
  const double f = jobsc[0], shift = jobsc[1], lin_inv_step = jobsc[2];
  // pixel terms of the model pass (first trip of its loop): under the backward sweep
  constexpr int PU = 6;
  double qlm[PU], qwk[PU];
  double2 qsg[PU];
  const bool cached = 2 * npix <= N;
  if (cached) {
#pragma unroll
    for (int u = 0; u < PU; u++) {
      const int k = min(tid + u * OBJ_NT, npix - 1);
      const double2 lp = AG.lp[k];   // (wavelength and knot coordinate: one request)
      qlm[u] = lp.x;
      qwk[u] = lp.y;
      qsg[u] = sig[k];
    }
  }
  {
    double z = 0, pb = 1;
#pragma unroll
    for (int q = OBJ_CHMAX - 1; q >= 0; q--)
      if (q < CH && a0 + q < a1) {
        const double ci = tfc[q];
        z = (loc[q] + pr[q] * d_in) - ci * z;   // d of the forward sweep, then z
        pb = -ci * pb;
        loc[q] = z;
        pr[q] = pb;
      }
    OBJ_T(13);   // (debug) backward recurrence
    const double z_in = chain3(z, pb, +1);
    OBJ_T(14);   // (debug) chunk hand-over
#pragma unroll
    for (int q = 0; q < OBJ_CHMAX; q++)
      if (q < CH && a0 + q < a1) dp[a0 + q] = loc[q] + pr[q] * z_in;
  }
  // ... and the knot terms of those pixels (their interval index needs the job's
  // velocity and the prefetched pixel terms only): in flight across the barrier
  double qkn[PU], qhk[PU], qik[PU];
  int qps[PU];
  if (cached) {
#pragma unroll
    for (int u = 0; u < PU; u++) {
      qlm[u] *= f;
      int pos = S.log_step ? (int)(qwk[u] + shift)
                           : (int)((qlm[u] - x0) * lin_inv_step);
      pos = min(max(pos, 0), N - 2);
      qps[u] = pos;
      // (knot i and i + 1 in one 16-byte request; h = their difference, the
      // subtraction the factors' own h came from: one request less per pixel)
      const double2 kk = knot_pair(pos);
      qkn[u] = kk.x;
      qhk[u] = kk.y - kk.x;
      qik[u] = ih[pos];
    }
  }
  __syncthreads();
  OBJ_T(3);
  // dp[u] = z at knot u+1; spline piece i in powers of dl = x - x_i exactly as
  // rvs_spline_construct(form 1) stores it
  auto tv_at = [&](int k) {
    const double x = AG.lam[k] * f;
    int pos = S.log_step ? (int)(AG.pix[k] + shift)
                         : (int)((x - x0) * lin_inv_step);
    pos = min(max(pos, 0), N - 2);
    const double dl = x - S.knots[pos];
    const double h = hh[pos], hinv = ih[pos];
    const double zi = (pos == 0) ? 0.0 : dp[pos - 1];
    const double zi1 = (pos + 1 == N - 1) ? 0.0 : dp[pos];
    const double yi = y[pos], yi1 = y[pos + 1];
    const double t1 = hinv * (1.0 / 6), t2 = h * (1.0 / 6);
    const double cb = (yi1 - yi) * hinv - t2 * (2 * zi + zi1);
    const double c2 = 0.5 * zi, c3 = (zi1 - zi) * t1;
    return fma(fma(fma(c3, dl, c2), dl, cb), dl, yi);
  };
  // ---- A10/A11: continuum-marginalised chi^2 (as point_block_kernel) -------
  double av[P];   // (the coefficients, from the Cholesky solve on)
  // model / data in units of sigma are parked in the (now free) factor buffer
  // when they fit: [npix] t_k/e_k, [npix] s_k/e_k.  They are produced by a
  // first pass that holds no accumulators, so it is unrolled over the pixels
  // of a thread with each round of loads issued together (pixel terms, then
  // the knot terms that depend on the interval index): two L2 round trips for
  // six pixels instead of two per pixel.
  double *tcache = bufC;
  if (cached) {
    constexpr int U = 6;
    for (int kb = tid; kb < npix; kb += U * OBJ_NT) {
      double lm[U], wk[U], kn[U], hk[U], ik[U];
      double2 sg[U];
      int ps[U];
      if (kb == tid) {
#pragma unroll
        for (int u = 0; u < U; u++) {
          lm[u] = qlm[u], sg[u] = qsg[u];
          ps[u] = qps[u], kn[u] = qkn[u], hk[u] = qhk[u], ik[u] = qik[u];
        }
      } else
      {
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int k = min(kb + u * OBJ_NT, npix - 1);
          const double2 lp = AG.lp[k];
          lm[u] = lp.x;
          wk[u] = lp.y;
          sg[u] = sig[k];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          lm[u] *= f;
          int pos = S.log_step ? (int)(wk[u] + shift)
                               : (int)((lm[u] - x0) * lin_inv_step);
          pos = min(max(pos, 0), N - 2);
          ps[u] = pos;
          const double2 kk = knot_pair(pos);
          kn[u] = kk.x;
          hk[u] = kk.y - kk.x;
          ik[u] = ih[pos];
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int k = kb + u * OBJ_NT, pos = ps[u];
        const double dl = lm[u] - kn[u];
        const double h = hk[u], hinv = ik[u];
        const double zi = (pos == 0) ? 0.0 : dp[pos - 1];
        const double zi1 = (pos + 1 == N - 1) ? 0.0 : dp[pos];
        const double yi = y[pos], yi1 = y[pos + 1];
        const double t1 = hinv * (1.0 / 6), t2 = h * (1.0 / 6);
        const double cb = (yi1 - yi) * hinv - t2 * (2 * zi + zi1);
        const double c2 = 0.5 * zi, c3 = (zi1 - zi) * t1;
        const double tv = fma(fma(fma(c3, dl, c2), dl, cb), dl, yi);
        if (k < npix) {
          tcache[k] = tv * sg[u].x;
          tcache[npix + k] = sg[u].y;
        }
      }
    }
  }
  OBJ_T(7);
  // Normal equations: rows [I0, I1) of the packed matrix + right-hand side, summed
  // over this thread's pixels, reduced over the wave by halving, wave totals into
  // red[w][].  One pass over the pixels up to P = 10 (65 sums: 130 registers); from
  // P = 11 the rows are split over two passes (objective_kernel<15> held 256 VGPRs
  // + 536 B of scratch per lane in round 3): a pass keeps about half of the sums, the
  // basis columns it needs are read again, every sum still receives its pixels in
  // the same order.
  auto te_dk_of = [&](int k, double &te, double &dk) {
    if (cached) {  // written by this same thread above
      te = tcache[k];
      dk = tcache[npix + k];
    } else {
      const double tv = tv_at(k);
      const double2 sg = sig[k];
      te = tv * sg.x;
      dk = sg.y;
    }
  };
  auto normal_pass = [&](auto i0_c, auto i1_c) {
    constexpr int I0 = decltype(i0_c)::value, I1 = decltype(i1_c)::value;
    constexpr int T0 = I0 * (I0 + 1) / 2, CM = I1 * (I1 + 1) / 2 - T0;
    constexpr int CNT = CM + (I1 - I0);
    double vals[CNT];   // [CM] matrix sums TRI(i, jj) - T0, then [I1 - I0] right-hand sides
#pragma unroll
    for (int i = 0; i < CNT; i++) vals[i] = 0;
    // basis rows one pixel ahead: the loads of the next pixel are in flight while
    // the FMAs of the current one issue
    if constexpr (P <= 14) {
      // (two row buffers taking turns: copied from one into the other at the head of
      // every pixel the row cost I1 register moves per pixel)
      double pn[I1], pm[I1];
      auto row = [&](double *dst, int k) {
        const double *pr = AG.polysT + (int64_t)min(k, npix - 1) * P;
#pragma unroll
        for (int i = 0; i < I1; i++) dst[i] = pr[i];
      };
      auto pixel = [&](const double *pv, int k) {
        double te, dk;
        te_dk_of(k, te, dk);
        const double wt = te * te, u = te * dk;
        double pw[I1];
#pragma unroll
        for (int i = 0; i < I1; i++) pw[i] = pv[i] * wt;
#pragma unroll
        for (int i = I0; i < I1; i++) {
          vals[CM + i - I0] = fma(pv[i], u, vals[CM + i - I0]);
#pragma unroll
          for (int jj = 0; jj <= i; jj++)
            vals[TRI(i, jj) - T0] = fma(pv[i], pw[jj], vals[TRI(i, jj) - T0]);
        }
      };
      row(pn, tid);
      for (int k = tid; k < npix;) {
        row(pm, k + OBJ_NT);
        pixel(pn, k);
        k += OBJ_NT;
        if (k >= npix) break;
        row(pn, k + OBJ_NT);
        pixel(pm, k);
        k += OBJ_NT;
      }
    } else {   // (15, 16 terms: no registers for a second row buffer)
      double pn[I1];
      {
        const double *pr = AG.polysT + (int64_t)min(tid, npix - 1) * P;
#pragma unroll
        for (int i = 0; i < I1; i++) pn[i] = pr[i];
      }
      for (int k = tid; k < npix; k += OBJ_NT) {
        double te, dk;
        te_dk_of(k, te, dk);
        const double wt = te * te, u = te * dk;
        double pv[I1], pw[I1];
#pragma unroll
        for (int i = 0; i < I1; i++) pv[i] = pn[i];
        {
          const double *pr = AG.polysT + (int64_t)min(k + OBJ_NT, npix - 1) * P;
#pragma unroll
          for (int i = 0; i < I1; i++) pn[i] = pr[i];
        }
#pragma unroll
        for (int i = 0; i < I1; i++) pw[i] = pv[i] * wt;
#pragma unroll
        for (int i = I0; i < I1; i++) {
          vals[CM + i - I0] = fma(pv[i], u, vals[CM + i - I0]);
#pragma unroll
          for (int jj = 0; jj <= i; jj++)
            vals[TRI(i, jj) - T0] = fma(pv[i], pw[jj], vals[TRI(i, jj) - T0]);
        }
      }
    }
    OBJ_T(4);
    // slot i of a lane stands for sum number base + i of this pass; lim = end of
    // the range that is really this lane's
    int cnt, base, lim;
    wave_halve<CNT>(vals, lane, cnt, base, lim);
#pragma unroll
    for (int i = 0; i < (CNT + 63) / 64 + 1; i++)
      if (i < cnt && base + i < lim) {
        const int a = base + i;
        red[w][a < CM ? T0 + a : NT + I0 + (a - CM)] = vals[i];
      }
  };
  {
    constexpr int PA = obj_split(P);
    // (red[] in the template buffer: every thread has left the model pass)
    if (RED_DYN) __syncthreads();
    normal_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, PA>{});
    if constexpr (PA < P)
      normal_pass(std::integral_constant<int, PA>{}, std::integral_constant<int, P>{});
  }
  __syncthreads();
  OBJ_T(8);   // (debug) wave reductions of the NV sums
  // fold of the waves' partial sums (wave order), one sum per thread
  if (tid < NV) {
    double v = red[0][tid];
#pragma unroll
    for (int q = 1; q < OBJ_NW; q++) v += red[q][tid];
    red[0][tid] = v;
  }
  __syncthreads();
  OBJ_T(9);   // (debug) fold over the waves
  // basis rows of the residual pass: requested before the Cholesky (one wave works
  // there, the other seven wait) instead of behind it
  // (6, 4, 3 or 2 pixels ahead at P = 10: no measurable difference)
  constexpr int RPF = (P <= 10) ? 6 : (P <= 12 ? 4 : 3);
  double qp[RPF][P];
  auto load_qp = [&]() {
#pragma unroll
    for (int u = 0; u < RPF; u++) {
      const double *prow = AG.polysT + (int64_t)min(tid + u * OBJ_NT, npix - 1) * P;
#pragma unroll
      for (int i = 0; i < P; i++) qp[u][i] = prow[i];
    }
  };
  // (wave 0 requests its rows behind the factorisation: 8 waves x 30 16-byte loads
  // take the CU's address pipe ~1.5 us to accept, and the chain everybody waits for
  // would start behind them)
  // The logarithms of L's diagonal (the determinant term) are off that chain too: wave 0
  // hands the diagonal over behind the factorisation and goes on with the back
  // substitution, the LAST wave -- idle at the barrier below, and the one with the fewest
  // pixels -- takes the logarithm of each element and sums them in the same order:
  // ~0.5 us of one wave's dependent fp64 instructions that every other wave waited for.
  // (An LDS flag, not a barrier: gfx950 has one barrier per block.  Wave 0 never waits
  // for the last wave before it raises the flag.)
  constexpr bool LOG_AWAY = OBJ_NW > 1;
  if (w != 0) load_qp();
  if (LOG_AWAY && w == OBJ_NW - 1) {
    while (__atomic_load_n(&dg_flag, __ATOMIC_RELAXED) == 0) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane < P) ldv[lane] = log(dgv[lane]);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      double ldet = 0;
#pragma unroll
      for (int q = 0; q < P; q++) ldet += ldv[q];
      coefs[P] = ldet;
    }
  }
  if (w == 0) {
    // Cholesky + the two triangular solves with ROW i on lane i (i < P):
    // left-looking, sums over q ascending as in the in-lane version of the
    // other chi^2 kernels, but the rows advance side by side: the serial chain
    // is P columns instead of P(P+1)/2 entries.  L is mirrored in LDS so that a
    // lane can read another row (same wave: LDS operations complete in order).
    // Lane P carries the right-hand side v as one more row of the matrix: what the
    // trailing updates leave there is y of L y = v -- v_k - sum_q L_kq y_q with q
    // ascending, times 1 / L_kk: the operations of the forward substitution that
    // followed the factorisation until round 6, now inside it.
    const int i = lane < P ? lane : P;
    double row[P];
#pragma unroll
    for (int jj = 0; jj < P; jj++) {
      const double m = red[0][i < P ? TRI(i, jj <= i ? jj : 0) : NT + jj];
      row[jj] = (jj <= i) ? m : 0.0;
    }
    bool ok = true;
    OBJ_TW(18);   // (debug) the rows out of LDS
    // dg / rdg: this lane's diagonal element of L and its reciprocal (the
    // off-diagonal elements and both triangular solves multiply by it: one
    // division per column on the serial chain).  A value of another row is a
    // register of another lane with a compile-time lane number: v_readlane,
    // no LDS round trip and no barrier on the chain.
    auto bcast = [](double v, int src) {
      const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
      const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
      return __hiloint2double(hi, lo);
    };
    double dg = 1.0, rdg = 1.0;
#pragma unroll
    for (int jj = 0; jj < P; jj++) {
      // right-looking: row[jj] already carries a_i,jj - sum_{q < jj} L_iq L_jj,q -- the
      // products of the left-looking form, subtracted in the same order (q ascending),
      // each as soon as column q existed: the trailing updates of a column are
      // independent of one another, and the serial chain of a column is the
      // reciprocal square root and one multiplication instead of jj dependent FMAs
      // behind it (45 in all at P = 10).  Same operations per entry: the same bits.
      const double sum = row[jj];
      // every lane runs the same instructions; lane jj's results are the ones
      // that count
      // d = sqrt(sum) and 1 / d from ONE reciprocal square root (hardware
      // estimate + two Newton steps: error ~1e-30 before rounding) instead of an
      // IEEE sqrt followed by an IEEE division -- some 45 dependent instructions
      // per column of a chain that the rest of the block waits for; d differs
      // from the correctly rounded root by at most an ulp
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
      Lm[i][jj] = row[jj];  // mirror for the back-substitution (column reads); row P: y
#pragma unroll
      for (int k = jj + 1; k < P; k++) row[k] -= row[jj] * bcast(row[jj], k);  // L[k][jj]
    }
    OBJ_TW(19);   // (debug) factorisation
    load_qp();
    OBJ_TW(20);   // (debug) wave 0's row requests
    // log of the diagonal: all rows at once (not one per column of the loop)
    if (LOG_AWAY) {
      if (lane < P) dgv[lane] = dg;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __atomic_store_n(&dg_flag, 1, __ATOMIC_RELAXED);
    } else if (lane < P) {
      ldv[lane] = log(dg);
    }
    OBJ_TW(21);   // (debug) log of the diagonal
    // L^T a = y from the last row up: a_ii from lane ii, the rows above subtract
    // L[ii][i] a_ii (column i of L and y_i out of the mirror, fetched ahead of the chain)
    __builtin_amdgcn_wave_barrier();
    const int ic = lane < P ? lane : P - 1;
    double col[P];
#pragma unroll
    for (int ii = 0; ii < P; ii++) col[ii] = Lm[ii][ic];
    double ti = Lm[P][ic];
#pragma unroll
    for (int ii = P - 1; ii >= 0; ii--) {
      const double aii = bcast(ti * rdg, ii);
      if (lane == 0) coefs[ii] = aii;
      if (ic < ii) ti -= col[ii] * aii;
    }
    const unsigned long long okm = __ballot(ok || lane >= P);
    if (lane == 0) {
      if (!LOG_AWAY) {
        double ldet = 0;
#pragma unroll
        for (int q = 0; q < P; q++) ldet += ldv[q];
        coefs[P] = ldet;
      }
      coefs[P + 1] = (okm == ~0ull) ? 1.0 : 0.0;
    }
    OBJ_TW(23);   // (debug) backward solve, determinant
  }
  __syncthreads();
  OBJ_T(5);
#pragma unroll
  for (int i = 0; i < P; i++) av[i] = coefs[i];
  double rr = 0;
  auto te_dk = [&](int k, double &te, double &dk) {
    if (cached) {
      te = tcache[k];
      dk = tcache[npix + k];
    } else {
      const double tv = tv_at(k);
      const double2 sg = sig[k];
      te = tv * sg.x;
      dk = sg.y;
    }
  };
  int kres = tid;
  // (the model / data values of those pixels: requested together, see the right-hand
  // sides above)
  double tev[RPF], dkv[RPF];
#pragma unroll
  for (int u = 0; u < RPF; u++)
    te_dk(min(tid + u * OBJ_NT, npix - 1), tev[u], dkv[u]);
#pragma unroll
  for (int u = 0; u < RPF; u++) {
    const int k = tid + u * OBJ_NT;
    if (k < npix) {
      const double te = tev[u], dk = dkv[u];
      double mdl = 0;
#pragma unroll
      for (int i = 0; i < P; i++) mdl = fma(av[i], qp[u][i], mdl);
      const double r = dk - mdl * te;
      rr = fma(r, r, rr);
    }
  }
  kres = tid + RPF * OBJ_NT;
  for (int k = kres; k < npix; k += OBJ_NT) {
    double te, dk;
    te_dk(k, te, dk);
    const double *pr = AG.polysT + (int64_t)k * P;
    double mdl = 0;
#pragma unroll
    for (int i = 0; i < P; i++) mdl = fma(av[i], pr[i], mdl);
    const double r = dk - mdl * te;
    rr = fma(r, r, rr);
  }
  OBJ_T(6);
  rr = wave_sum(rr);
  __syncthreads();
  if (lane == 0) red[w][0] = rr;
  __syncthreads();
  if (tid == 0) {
    rr = red[0][0];
    for (int q = 1; q < OBJ_NW; q++) rr += red[q][0];
    const double lz = AG.wbase[2ll * S.S * npix + 2 * s];
    double chi = 2.0 * coefs[P] + 2.0 * lz + rr;
    int st = st_extra;
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
    armout[(int64_t)blockIdx.y * J + j] = outside;
  }
}

// arms in order; penalties of A11 (spec_fit.py:888-896)
__global__ void objective_sum_kernel(int narm, int J, double badchi,
                                     const double *__restrict__ pen_scale,
                                     const int32_t *__restrict__ job_spec,
                                     const int32_t *__restrict__ live,
                                     int outside_penalty,
                                     const double *__restrict__ armchi,
                                     const int32_t *__restrict__ armst,
                                     const double *__restrict__ armout,
                                     double *__restrict__ out,
                                     int32_t *__restrict__ status) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= J || (live && j >= live[0])) return;
  // (badchi = 10 x the pixel count of the spectrum, spec_fit.py:863: the factor
  // of a spectrum on a shorter grid of a grid set)
  if (pen_scale) badchi *= pen_scale[job_spec ? job_spec[j] : j];
  ObjArmOut AO;
  AO.armchi = armchi, AO.armout = armout, AO.armst = armst, AO.narm = narm, AO.J = J;
  double tot;
  int st;
  obj_sum_row(AO, j, badchi, outside_penalty, tot, st);
  out[j] = tot;
  if (outside_penalty & 2)
    status[j] = st;  // RVS_OBJ_STATUS_STORE: the caller's buffer is scratch
  else if (st)
    atomicOr(&status[j], st);
}

// largest template grid (knots) the kernel can hold in LDS for this npoly:
// 160 KB per workgroup minus the kernel's static LDS, three doubles per knot
extern "C" int rvs_objective_max_ntp(int npoly) {
  hipFuncAttributes at;
  const void *fn = nullptr;
#define RVS_CASE(PP)                               \
  case PP:                                         \
    fn = (const void *)objective_kernel<PP, false>;\
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
  const int64_t room = 160 * 1024 - (int64_t)at.sharedSizeBytes -
                       2 * OBJ_FIR_PAD * (int64_t)sizeof(double);
  int n = (int)(room / (3 * (int64_t)sizeof(double)));
  return n > 8192 ? 8192 : (n < 0 ? 0 : n);
}

extern "C" int64_t rvs_objective_work_size(int J, int narm) {
  if (J < 1 || narm < 1) return 0;
  // per (arm, job): chi^2, outside, status (padded to 8 bytes) + the cell-search
  // record of objective_locate_kernel
  // ... + the jobs' order (objective_order_kernel)
  return (int64_t)narm * J *
             (int64_t)(3 * sizeof(double) + OBJ_LOC_REC * sizeof(double)) +
         8 * (((int64_t)J + 1) / 2);
}

static int objective_launch(const rvs_objective_arm *arms, int narm, int npoly,
                            const ObjTempl *tt, const double *params,
                            const double *vsini, const int32_t *job_spec, int J,
                            const double *vel, double badchi,
                            int outside_penalty, void *scratch, double *out,
                            int32_t *status, const int32_t *live, void *stream) {
  if (J < 1 || narm < 1 || narm > RVS_MAX_ARMS || !arms || !scratch)
    return RVS_E_ARG;
  ObjArms A;
  A.n = narm;
  ObjTempl TT = {};
  if (tt) TT = *tt;
  size_t shm = 0;
  for (int i = 0; i < narm; i++) {
    A.a[i] = arms[i];
    // (ntp >= 32: the FIR window and the spline chunks read up to 24 doubles behind a
    // thread's own rows without testing, inside the block's three template buffers)
    if (arms[i].pt.npix < 1 || arms[i].ntp < 32 || arms[i].ntp > 8192 ||
        arms[i].pt.taps || arms[i].pt.fast_interp || !arms[i].factors)
      return RVS_E_ARG;
    if (tt) {
      if (!TT.templ[i] || !TT.outside[i]) return RVS_E_ARG;
    } else if (arms[i].ndim < 1 || arms[i].ndim > MAXDIM) {
      return RVS_E_ARG;
    }
    shm = max(shm, (size_t)(3 * arms[i].ntp + 2 * OBJ_FIR_PAD) * sizeof(double));
  }
  for (int i = narm; i < RVS_MAX_ARMS; i++) {
    A.a[i] = arms[0];
    TT.templ[i] = TT.templ[0];
    TT.outside[i] = TT.outside[0];
  }
  hipStream_t st = rvs_stream(stream);
  double *armchi = (double *)scratch;
  double *armout = armchi + (int64_t)narm * J;
  int32_t *armst = (int32_t *)(armout + (int64_t)narm * J);
  double *locbuf = armout + 2 * (int64_t)narm * J;   // (status padded to 8 B)
  dim3 grid(J, narm);
  if (shm > (size_t)(3 * rvs_objective_max_ntp(npoly) + 2 * OBJ_FIR_PAD) * sizeof(double))
    return RVS_E_ARG;
  if (npoly > 10)   // (the wave totals share the template buffer: RED_DYN)
    for (int i = 0; i < narm; i++)
      // (... which the model pass has left: template and second derivatives, two
      // buffers of ntp doubles, for OBJ_NW rows of NV + 1 sums)
      if (2 * arms[i].pt.npix > arms[i].ntp ||
          2 * arms[i].ntp < OBJ_NW * (npoly * (npoly + 1) / 2 + npoly + 1))
        return RVS_E_ARG;
  const double *loc = nullptr;
  int32_t *perm = nullptr;
  if (!tt) {
    bool pre = true;
    for (int i = 0; i < narm; i++)
      if ((1 << arms[i].ndim) > OBJ_LOC_NV) pre = false;
    // a launch of up to three passes over the CUs (<= 768 blocks; 256 until round 6:
    // --process 500 1498 -> 1548 spectra/s, 2000 equal, 3072 slower -- the launches
    // lose their cell order): the cell search inside
    // the block -- the same device function, so the same values -- and one launch
    // less per evaluation in the optimiser's latency-bound last rounds (+1 % at 2000
    // spectra; option obj_inblk_max overrides the bound, 0 = never)
    if ((int64_t)J * narm <= rvs_opt(RVS_OPT_OBJ_INBLK_MAX)) pre = false;
    if (pre) {
      hipLaunchKernelGGL(objective_locate_kernel, grid, dim3(OBJ_LOC_NT), 0, st, A,
                         params, J, live, vel, vsini, 0.6, locbuf);
      loc = locbuf;
      // from a few blocks per CU up: jobs in cell order (obj_sort = 0: a test
      // hook, tests/test_gpu_parity.py::test_objective_job_order)
      if (J >= OBJ_SORT_MIN_JOBS && rvs_opt(RVS_OPT_OBJ_SORT)) {
        int shift = 0;
        while (((arms[0].ngrid - 1) >> shift) >= OBJ_ORD_NB) shift++;
        const int nb = (int)((arms[0].ngrid - 1) >> shift) + 1;
        perm = (int32_t *)(locbuf + (int64_t)narm * J * OBJ_LOC_REC);
        hipLaunchKernelGGL(objective_order_kernel, dim3(1), dim3(OBJ_ORD_NT), 0, st,
                           locbuf, J, live, shift, nb, perm);
      }
    }
  }
#ifdef OBJ_PIPE_EXPERIMENT
  // tools/perf/experiments/objective_pipe.hip (obj_bench only): the persistent
  // producer/consumer kernel where it applies; RVS_OBJ_PIPE=0 keeps this file's
  {
    static const bool use_pipe = [] {   // (obj_bench only: read once)
      const char *ev = getenv("RVS_OBJ_PIPE");
      return !(ev && ev[0] == '0');
    }();
    if (use_pipe && (tt || loc)) {
      const int prc = objective_pipe_launch(A, tt ? &TT : nullptr, npoly, loc, vsini,
                                            job_spec, J, vel, (shm - 2 * OBJ_FIR_PAD * sizeof(double)) / (3 * sizeof(double)),
                                            armchi, armst, armout, st);
      if (prc == 0) {
        hipLaunchKernelGGL(objective_sum_kernel, dim3((J + 255) / 256), dim3(256), 0,
                           st, narm, J, badchi, arms[0].pt.pen_scale, job_spec, live,
                           outside_penalty, armchi, armst,
                           armout, out, status);
        RVS_LAUNCH_CHECK();
        return 0;
      }
      if (prc != RVS_E_ARG) return prc;
    }
  }
#endif
#define RVS_LAUNCH_OBJ(PP, FT, IB)                                                 \
  {                                                                            \
    static bool attr_set = false;                                              \
    if (!attr_set) {                                                           \
      (void)hipFuncSetAttribute((const void *)objective_kernel<PP, FT, IB>,    \
                                hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                160 * 1024 - 1024);                            \
      (void)hipGetLastError();                                                 \
      attr_set = true;                                                         \
    }                                                                          \
    hipLaunchKernelGGL((objective_kernel<PP, FT, IB>), grid, dim3(OBJ_NT), shm,  \
                       st,                                                     \
                       A, TT, loc, perm, live, params, vsini, job_spec, J, vel, 0.6, \
                       armchi, armst, armout);                                 \
  }
#define RVS_CASE(PP)                                                           \
  case PP:                                                                     \
    if (tt) RVS_LAUNCH_OBJ(PP, true, false)                                    \
    else if (loc) RVS_LAUNCH_OBJ(PP, false, false)                             \
    else RVS_LAUNCH_OBJ(PP, false, true)                                       \
    break;
  switch (npoly) {
    RVS_ALL_CASES
    default:
      return RVS_E_ARG;
  }
#undef RVS_CASE
#undef RVS_LAUNCH_OBJ
  if (outside_penalty & RVS_OBJ_NO_SUM) {   // the caller sums the arms (nm.hip)
    RVS_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(objective_sum_kernel, dim3((J + 255) / 256), dim3(256), 0,
                     st, narm, J, badchi, arms[0].pt.pen_scale, job_spec, live,
                           outside_penalty, armchi, armst, armout,
                     out, status);
  RVS_LAUNCH_CHECK();
  return 0;
}

extern "C" int rvs_objective_fused(const rvs_objective_arm *arms, int narm,
                                   int npoly, const double *params,
                                   const double *vsini, const int32_t *job_spec,
                                   int J, const double *vel, double badchi,
                                   int outside_penalty, void *scratch,
                                   double *out, int32_t *status, void *stream) {
  return objective_launch(arms, narm, npoly, nullptr, params, vsini, job_spec, J,
                          vel, badchi, outside_penalty, scratch, out, status,
                          nullptr, stream);
}

extern "C" int rvs_objective_fused_n(const rvs_objective_arm *arms, int narm,
                                     int npoly, const double *params,
                                     const double *vsini, const int32_t *job_spec,
                                     int J, const int32_t *njobs_dev,
                                     const double *vel, double badchi,
                                     int outside_penalty, void *scratch,
                                     double *out, int32_t *status, void *stream) {
  return objective_launch(arms, narm, npoly, nullptr, params, vsini, job_spec, J,
                          vel, badchi, outside_penalty, scratch, out, status,
                          njobs_dev, stream);
}

extern "C" int rvs_objective_from_template(
    const rvs_objective_arm *arms, int narm, int npoly,
    const double *const *templ, const double *const *outside,
    const double *vsini, const int32_t *job_spec, int J, const double *vel,
    double badchi, int outside_penalty, void *scratch, double *out,
    int32_t *status, void *stream) {
  return rvs_objective_from_template_n(arms, narm, npoly, templ, outside, vsini,
                                       job_spec, J, nullptr, vel, badchi,
                                       outside_penalty, scratch, out, status,
                                       stream);
}

extern "C" int rvs_objective_from_template_n(
    const rvs_objective_arm *arms, int narm, int npoly,
    const double *const *templ, const double *const *outside,
    const double *vsini, const int32_t *job_spec, int J,
    const int32_t *njobs_dev, const double *vel, double badchi,
    int outside_penalty, void *scratch, double *out, int32_t *status,
    void *stream) {
  if (!templ || !outside || narm < 1 || narm > RVS_MAX_ARMS) return RVS_E_ARG;
  ObjTempl tt = {};
  for (int i = 0; i < narm; i++) {
    tt.templ[i] = templ[i];
    tt.outside[i] = outside[i];
  }
  return objective_launch(arms, narm, npoly, &tt, nullptr, vsini, job_spec, J,
                          vel, badchi, outside_penalty, scratch, out, status,
                          njobs_dev, stream);
}
