// NeuS-facto BASELINE config 2/3: 8x256 geometry MLP (skip at 4), in0 = 71 (16x2 grid), 4x256 colour MLP.
#include "field_inst.h"
SDFHIP_DEFINE_FIELD_KERNELS(A, 8, 3, 6, 8, 4, 8, 3, 8, 4)
