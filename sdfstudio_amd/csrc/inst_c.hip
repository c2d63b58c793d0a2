// neus-facto-angelo field widths (BASELINE config 5, method_configs.py:403-422): 256-wide geometry MLP on
// in0 = 3 + 36 (zeroed PE) + 16 x 8 grid features = 167 (6 blocks), 256-wide colour MLP with appearance embedding; any depth
// (the preset: 1 hidden geometry layer, no skip, 4 colour layers).
// This unit: colour network kernels and the kernel table; the geometry kernels are in inst_c_fwd.hip / inst_c_inf.hip / inst_c_bwd.hip.
#include "field_inst.h"
SDFHIP_DEFINE_COL_AND_TABLE_(C, 8, 6, 8, 3, 8, 1)
