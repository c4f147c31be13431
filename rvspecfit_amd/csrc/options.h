// The behaviour switches of librvsgpu (options.cpp; include/rvsgpu.h documents
// them under rvs_option_set).
#pragma once
#include "../../include/rvsgpu.h"

enum {
  RVS_OPT_XC_WS,          // wave-specialised cross-correlation where it applies
  RVS_OPT_XC_WS1,         // nfft 4096: one template per iteration instead of two
  RVS_OPT_NM_GLUE,        // optimiser rounds as three bookkeeping kernels
  RVS_OPT_NM_BUCKET,      // launch bounds rounded to buckets (measurement hook)
  RVS_OPT_OBJ_INBLK_MAX,  // launches of <= this many blocks search their cell in-block
  RVS_OPT_OBJ_SORT,       // objective jobs in grid-cell order
  RVS_OPT_NN_PIPE,        // MLP wide last layer: the kernel with the pipelined epilogue
  RVS_OPT_NM_SPLIT_MIN,   // rounds of >= this many rows: row-parallel bookkeeping + pack
  RVS_OPT_NM_SPEC_MAX,    // rounds of <= this many rows: all candidates in one launch
  RVS_OPT_NM_TAIL_WINDOW, // rounds between two looks of the host once <= 256 rows are live
  RVS_OPT_COUNT
};
// current value (the table is filled from the environment on first use)
int rvs_opt(int id);
