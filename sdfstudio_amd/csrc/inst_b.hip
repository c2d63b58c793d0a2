// Small parity configuration (tests/golden): 8x64 geometry MLP (skip at 4), in0 = 55 (8x2 grid), 4x64 colour MLP.
#include "field_inst.h"
SDFHIP_DEFINE_FIELD_KERNELS(B, 2, 2, 2, 8, 4, 2, 3, 2, 4)
