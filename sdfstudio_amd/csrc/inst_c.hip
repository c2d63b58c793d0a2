// neus-facto-angelo field shape (BASELINE config 5, method_configs.py:403-422): 1-hidden-layer 256-wide geometry MLP on
// in0 = 3 + 36 (zeroed PE) + 16 x 8 grid features = 167 (6 blocks), no skip connection, 4x256 colour MLP with appearance embedding.
#include "field_inst.h"
SDFHIP_DEFINE_FIELD_KERNELS(C, 8, 6, 0, 1, -1, 8, 3, 8, 4)
