// 256-wide geometry MLP on in0 = 71 (3 + 36 PE + 16 x 2 grid features: 3 blocks), 256-wide geometry feature, 256-wide colour MLP:
// BASELINE configs 1 - 4 (8 layers, skip at 4, 4 colour layers), the reference's neus-facto preset (2 + 2 layers,
// method_configs.py:474-476) and every other depth of the same widths.
// This unit: colour network kernels and the kernel table; the geometry kernels are in inst_a_fwd.hip / inst_a_inf.hip / inst_a_bwd.hip.
#include "field_inst.h"
SDFHIP_DEFINE_COL_AND_TABLE(A, 8, 3, 8, 3, 8)
