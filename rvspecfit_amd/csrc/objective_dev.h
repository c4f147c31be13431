// Declarations shared by objective.hip (the per-block kernel and the entry
// points) and objective_pipe.hip (the persistent producer/consumer kernel).
#pragma once
#include "template_dev.h"

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
#define OBJ_LOC_REC (2 * OBJ_LOC_NV + 2)   // doubles: w[16], id[16], dist, {mode, nearest}

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
