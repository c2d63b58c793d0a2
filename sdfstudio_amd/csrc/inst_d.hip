// The `neus-facto` PRESET's field shape (method_configs.py:472-480: num_layers = 2, num_layers_color = 2, hidden_dim = 256):
// 2x256 geometry MLP without skip connection on in0 = 71 (16 x 2 grid), 2x256 colour MLP.
#include "field_inst.h"
SDFHIP_DEFINE_FIELD_KERNELS(D, 8, 3, 0, 2, -1, 8, 3, 8, 2)
