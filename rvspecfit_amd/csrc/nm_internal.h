// Internal (not exported) entry of nm.hip shared with bfgs_dev.hip.
#pragma once
#include "common.h"

// F[j] = chisq_func(X[j]) for spectrum list[j], j < min(J, counts[cidx]) (counts
// == NULL: all J): rvs_proc_map -> the objective (rvs_objective_fused_n, or
// rvs_template_nn_arms_n + rvs_objective_from_template_n for MLP libraries) ->
// rvs_proc_finish, on the row buffers of `o` (capacity >= J rows).
// template.hip: the rows of a round through the Delaunay evaluators of all arms
__attribute__((visibility("hidden"))) int rvs_internal_template_tri_arms_n(
    const double *params, int B, const int32_t *live, int ndim, int narm,
    const rvs_nm_tri_arm *arms, hipStream_t st);

__attribute__((visibility("hidden"))) int rvs_internal_nm_eval(
    const rvs_nm_objective *o, const int32_t *list, const double *X, int J,
    const int32_t *counts, int cidx, double *F, hipStream_t st);
