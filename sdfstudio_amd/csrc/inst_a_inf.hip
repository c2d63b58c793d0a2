#include "field_inst.h"

SDFHIP_DEFINE_GEO_FWD_INFER_PAIR(A, 8, 3, 8)  // + the pair-wave form of the sdf-only forward (pair_kernels.h)
