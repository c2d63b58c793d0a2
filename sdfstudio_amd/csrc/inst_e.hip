// TCNNNerfactoField, the "grid" background model of BASELINE config 5 (fields/nerfacto_field.py:128-156, 211-225): two bias-free
// 64-wide ReLU MLPs (tcnn FullyFusedMLP in the reference) - base: 16 x 2 hash-grid features (in0 = 3 + 32: 2 blocks) -> 64 ->
// density + 15 features (one 32-wide feature block); head: [features | spherical harmonics (16) + appearance embedding (32), carried
// in the per-ray embedding slots of the small-input blocks (3 blocks)] -> 64 -> 64 -> rgb.  First-order kernels only.
#include "field_inst.h"
SDFHIP_DEFINE_FIRST_ORDER_FIELD_KERNELS(E, 2, 2, 1, 3, 2)
