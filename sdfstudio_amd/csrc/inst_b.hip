// 64-wide networks on in0 = 55 (2 blocks): the small configuration of the golden vectors (tests/golden), any depth.
#include "field_inst.h"
SDFHIP_DEFINE_FIELD_KERNELS(B, 2, 2, 2, 3, 2)
