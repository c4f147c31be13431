// Declarations shared by objective.hip (the per-block kernel and the entry
// points) and objective_pipe.hip (the persistent producer/consumer kernel).
#pragma once
#include "template_dev.h"
#include "objective_sum.h"

#define TRI(i, j) ((i) * ((i) + 1) / 2 + (j))

// (tools/perf/obj_bench.hip builds one npoly only: -DOBJ_ONLY_P=10)
#ifdef OBJ_ONLY_P
#define RVS_ALL_CASES RVS_CASE(OBJ_ONLY_P)
#else
#define RVS_ALL_CASES                                                     \
  RVS_CASE(1) RVS_CASE(2) RVS_CASE(3) RVS_CASE(4) RVS_CASE(5) RVS_CASE(6) \
  RVS_CASE(7) RVS_CASE(8) RVS_CASE(9) RVS_CASE(10) RVS_CASE(11)           \
  RVS_CASE(12) RVS_CASE(13) RVS_CASE(14) RVS_CASE(15) RVS_CASE(16)
#endif

// the grid of spectrum s on an arm (rvs_point_arm.grid_id, include/rvsgpu.h):
// wavelengths, pixel knot coordinates, basis, base of the per-spectrum blocks of
// the rvs_chisq_prepare buffer
struct ObjArmGrid {
  const double *lam, *pix, *polysT, *wbase;
  const double2 *lp;   // {lam, pix} of the grid's pixels (rvs_chisq_prepare)
};
__device__ __forceinline__ ObjArmGrid obj_arm_grid(const rvs_point_arm &T, int s) {
  ObjArmGrid g;
  const int G = T.G > 1 ? T.G : 1;
  const int64_t gi = (T.G > 1 && T.grid_id) ? T.grid_id[s] : 0;
  g.lam = T.lam + gi * T.npix;
  g.pix = T.work + gi * T.npix;
  g.polysT = T.polysT + gi * T.polys_stride;
  g.wbase = T.work + (int64_t)G * T.npix;
  g.lp = reinterpret_cast<const double2 *>(g.wbase + 4ll * T.S * T.npix + 2ll * T.S) +
         gi * T.npix;
  return g;
}

struct ObjArms {
  rvs_objective_arm a[RVS_MAX_ARMS];
  int n;
};

// FROMT: the unbroadened template of every (job, arm) comes from HBM -- a row
// of an evaluator that is no grid gather (the MLP of rvs_template_nn) -- with
// its outside flag; everything behind the template (FIR, spline solve, chi^2)
// is the same code.
struct ObjTempl {
  const double *templ[RVS_MAX_ARMS];    // [J, ntp] per arm
  const double *outside[RVS_MAX_ARMS];  // [J] per arm
};

// The cell search of a (job, arm) -- log10 mapping, one binary search per
// dimension, 2^ndim idgrid look-ups, weights; outside the grid the brute-force
// nearest neighbour -- is a chain of dependent loads that a few threads walk while
// the other 500 of an objective block wait: 7.7 % of the block's time on the one
// block a CU can hold.  objective_locate_kernel runs it ahead for all (job, arm)
// pairs of a launch with one wave each (thousands of them resident at once, their
// latencies overlapping) and leaves a record the objective block fetches with one
// coalesced load.  Same poly_locate code: the same ids, weights and distances.
#define OBJ_LOC_NV 16                      // grids of up to 4 dimensions
// doubles: w[16], id[16], dist, {mode, nearest}, the job's Doppler scalars {f, shift,
// 1 / linear step} (obj_job_scalars), the normalised taps 0 .. kmax of a rotational
// kernel of half width kmax <= OBJ_FIR_KMAX (obj_rot_taps)
#ifndef OBJ_FIR_KMAX
#define OBJ_FIR_KMAX 8   // widest rotational kernel (half width) of the register-window FIR
#endif
#define OBJ_LOC_ROT (2 * OBJ_LOC_NV + 5)    // {kmax, refused} of the rotational kernel
#define OBJ_LOC_TAPS (2 * OBJ_LOC_NV + 6)
#define OBJ_LOC_REC (OBJ_LOC_TAPS + OBJ_FIR_KMAX + 1)

// half width of a job's rotational kernel on the arm's template grid: 0 = none (no
// rotation, or refused: `refused`), R = v sin i / c in template pixels
__device__ __forceinline__ int obj_rot_kmax(const rvs_objective_arm &T, double vs,
                                            double &R, bool &refused) {
  R = (vs / RVS_C_KMS) / T.lnstep;
  refused = false;
  if (!(vs > 0) || (R < 1e-9)) return 0;
  const int kmax = (int)ceil(R + 1);
  // (the kernel's primitives are staged in a buffer of the template's length as two
  // runs of kmax + 3 doubles: a kernel wider than that is refused like one wider than
  // the template)
  if (kmax >= T.ntp || 2 * (kmax + 3) > T.ntp) {
    refused = true;
    return 0;
  }
  return kmax;
}
// tap k (not yet normalised) from the primitives at the clipped points x_j = clip(j /
// R), j = -1 .. kmax + 1, staged as entries j + 1 of pk0 / pk1 (rot_prim)
__device__ __forceinline__ double obj_rot_tap_raw(int k, double R, const double *pk0,
                                                  const double *pk1) {
  double ww = 0;
  // x_{k-1}, x_k, x_{k+1} are entries k, k+1, k+2
  double lo = fmin(fmax(k / R, -1.0), 1.0), hi = fmin(fmax((k + 1) / R, -1.0), 1.0);
  if (hi > lo)   // rot_segment(lo, hi, -R, 1 + k)
    ww += -R * (pk1[k + 2] - pk1[k + 1]) + (1.0 + k) * (pk0[k + 2] - pk0[k + 1]);
  lo = fmin(fmax((k - 1) / R, -1.0), 1.0);
  hi = fmin(fmax(k / R, -1.0), 1.0);
  if (hi > lo)   // rot_segment(lo, hi, R, 1 - k)
    ww += R * (pk1[k + 1] - pk1[k]) + (1.0 - k) * (pk0[k + 1] - pk0[k]);
  return ww;
}

// What a job's velocity turns into, the same for every pixel of the (job, arm): the
// Doppler factor f = sqrt((1 - b) / (1 + b)) (spec_fit.py:707-727), the pixels' shift in
// knot coordinates on a log-uniform grid, ln f / ln(knot ratio), or the inverse step of a
// linear one.  Two divisions, a square root and two logarithms -- ~330 instructions that
// all 512 threads of an objective block used to execute for themselves (1.3 us of the
// block's VALU issue): now ONE lane, of the cell-search kernel where it runs ahead
// (with the cell record), else of a wave that idles under the rotational kernel's
// construction.  Same expressions: same values.
__device__ __forceinline__ void obj_job_scalars(const rvs_point_arm &S, double velj,
                                                double *out) {
  const double bb = velj / RVS_C_KMS;
  const double f = sqrt((1.0 - bb) / (1.0 + bb));
  const double x0 = S.knots[0];
  out[0] = f;
  out[1] = S.log_step ? log(f) / log(S.knots[1] / x0) : 0.0;
  out[2] = S.log_step ? 0.0 : 1.0 / (S.knots[1] - x0);
}

__device__ __forceinline__ GridDesc obj_grid_desc(const rvs_objective_arm &T) {
  GridDesc G;
  const int nd = T.ndim;
  G.ndim = nd;
  G.log_mask = T.log_mask;
  int off = 0;
#pragma unroll
  for (int d = 0; d < MAXDIM; d++) {   // (static indices: the descriptor stays in registers)
    G.lens[d] = (d < nd) ? T.lens[d] : 1;
    G.uoff[d] = off;
    off += G.lens[d];
    G.ptp[d] = (d < nd) ? T.ptp[d] : 1.0;
  }
  int64_t st = 1;
#pragma unroll
  for (int d = MAXDIM - 1; d >= 0; d--) {
    if (d < nd) {
      G.gstride[d] = st;
      st *= T.lens[d];
    } else {
      G.gstride[d] = 0;
    }
  }
  return G;
}


// Wave reduction of CNT per-lane sums by halving: at the step with lane mask m the
// lanes l and l^m split the live sums between them (the lane with the bit clear
// keeps the lower half), so a step moves half as many values as the one before:
// ~CNT exchanges in all instead of CNT full butterflies.  The exchange at mask 32
// / 16 -- the upper half (odd rows) of the kept value trades places with the lower
// half (even rows) of the sent one -- IS v_permlane32_swap / v_permlane16_swap
// (gfx950), on the VALU; masks 8..1 are DPP row permutations.  (As ds_bpermute the
// exchanges of 8 waves queued on the CU's one LDS crossbar: 13.8 % of the block.)
// After the six steps slot i (< cnt) of a lane holds the wave total of sum number
// base + i when that number is < lim.
template <int CNT>
__device__ __forceinline__ void wave_halve(double (&vals)[CNT], int lane,
                                           int &cnt_o, int &base_o, int &lim_o) {
  int cnt = CNT, base = 0, lim = CNT;
#pragma unroll
  for (int mk = 32; mk >= 1; mk >>= 1) {
    const int h = (cnt + 1) >> 1;
    const bool up = (lane & mk) != 0;
#pragma unroll
    for (int i = 0; i < h; i++) {
      const bool has_hi = (i + h < cnt);
      const double lo = vals[i], hi = has_hi ? vals[i + h] : 0.0;
      if (mk >= 16) {
        const unsigned l0 = __double2loint(lo), l1 = __double2hiint(lo);
        const unsigned h0 = __double2loint(hi), h1 = __double2hiint(hi);
        if (mk == 32) {
          const auto r0 = __builtin_amdgcn_permlane32_swap(l0, h0, false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(l1, h1, false, false);
          vals[i] = __hiloint2double(r1[0], r0[0]) + __hiloint2double(r1[1], r0[1]);
        } else {
          const auto r0 = __builtin_amdgcn_permlane16_swap(l0, h0, false, false);
          const auto r1 = __builtin_amdgcn_permlane16_swap(l1, h1, false, false);
          vals[i] = __hiloint2double(r1[0], r0[0]) + __hiloint2double(r1[1], r0[1]);
        }
      } else {
        const double send = up ? lo : hi;
        const double keep = up ? hi : lo;
        double recv;
        if (mk == 8)
          recv = dpp_get<0x141, 0xf, 0xf>(dpp_get<0x140, 0xf, 0xf>(send));
        else if (mk == 4)
          recv = dpp_get<0x1b, 0xf, 0xf>(dpp_get<0x141, 0xf, 0xf>(send));
        else if (mk == 2)
          recv = dpp_get<0x4e, 0xf, 0xf>(send);
        else
          recv = dpp_get<0xb1, 0xf, 0xf>(send);
        vals[i] = keep + recv;
      }
    }
    lim = up ? lim : min(lim, base + h);
    base += up ? h : 0;
    cnt = h;
  }
  cnt_o = cnt;
  base_o = base;
  lim_o = lim;
}


// rows [0, obj_split(P)) of the normal equations are summed in the first pass over
// the pixels: everything up to P = 10, about half of the P (P + 3) / 2 sums beyond
#ifndef OBJ_ONEPASS_MAX
#define OBJ_ONEPASS_MAX 10
#endif
__host__ __device__ constexpr int obj_split(int P) {
  if (P <= OBJ_ONEPASS_MAX) return P;
  int pa = 1;
  while (pa < P && pa * (pa + 3) < P * (P + 3) / 2) pa++;
  return pa;
}

#ifdef OBJ_PIPE_EXPERIMENT
// tools/perf/experiments/objective_pipe.hip: the persistent producer/consumer
// kernel.  Returns RVS_E_ARG when the launch is outside what it covers.
int objective_pipe_launch(const ObjArms &A, const ObjTempl *tt, int npoly,
                          const double *locrec, const double *vsini,
                          const int32_t *job_spec, int J, const double *vel,
                          size_t nmax, double *armchi, int32_t *armst,
                          double *armout, hipStream_t st);
int objective_pipe_max_ntp(int npoly);
#endif
