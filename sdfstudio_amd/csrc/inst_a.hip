// NeuS-facto BASELINE config 2/3: 8x256 geometry MLP (skip at 4), in0 = 71 (16x2 grid), 4x256 colour MLP.
// This unit: colour network kernels and the kernel table; the geometry kernels are in inst_a_fwd.hip / inst_a_bwd.hip.
#include "field_inst.h"
SDFHIP_DEFINE_COL_AND_TABLE(A, 8, 3, 8, 8, 4, 8, 3, 8, 4)
