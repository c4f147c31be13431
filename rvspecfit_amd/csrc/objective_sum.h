// The per-arm results of an objective launch and their sum over the arms: shared by
// objective.hip (objective_sum_kernel) and nm.hip (the optimiser's round kernels).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

// Per-arm results of an objective launch in the caller's scratch (objective_launch):
// [narm, J] chi^2, [narm, J] outside flag, [narm, J] int32 status -- J the launch's
// job bound.  Their sum over the arms is objective_sum_kernel's work; the lock-step
// optimiser's own kernels (nm.hip) do it inline instead of waiting for one more launch.
struct ObjArmOut {
  const double *armchi, *armout;
  const int32_t *armst;
  int32_t narm;
  int64_t J;
};
__host__ __device__ inline ObjArmOut obj_arm_out(const void *scratch, int narm, int J) {
  ObjArmOut A;
  A.armchi = (const double *)scratch;
  A.armout = A.armchi + (int64_t)narm * J;
  A.armst = (const int32_t *)(A.armout + (int64_t)narm * J);
  A.narm = narm;
  A.J = J;
  return A;
}
// out[j] and status[j] of rvs_objective_fused for job j (badchi already scaled for
// the job's spectrum): spec_fit.py:888-896
__device__ __forceinline__ void obj_sum_row(const ObjArmOut &A, int j, double badchi,
                                            int outside_penalty, double &tot_o,
                                            int &st_o) {
#pragma clang fp contract(fast)   // (as objective.hip is compiled: one arithmetic)
  double tot = 0;
  int st = 0;
  for (int ia = 0; ia < A.narm; ia++) {
    const double o = A.armout[(int64_t)ia * A.J + j];
    if (!(fabs(o) <= 1.79e308)) {
      tot += 1000.0 * badchi;
      continue;
    }
    tot += A.armchi[(int64_t)ia * A.J + j] +
           ((outside_penalty & 1) ? o * badchi : 0.0);
    st |= A.armst[(int64_t)ia * A.J + j];
  }
  tot_o = tot;
  st_o = st;
}

