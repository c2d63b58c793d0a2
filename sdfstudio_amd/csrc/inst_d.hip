// NeRFField, the reference's default background model (background_model = "mlp", base_surface_model.py:189-200;
// fields/vanilla_nerf_field.py:37-114): ReLU geometry-type network 256 wide on in0 = 63 (x + 10-frequency encoding: 2 blocks), any
// depth (the reference: 8 layers, skip at 4), 128-wide "feature" = the base-output columns of the head MLP's first layer, colour-type
// head 128 wide on [feature | direction encoding] (2 small-input blocks).  First-order kernels only (field_inst.h).
#include "field_inst.h"
SDFHIP_DEFINE_FIRST_ORDER_FIELD_KERNELS(D, 8, 2, 4, 2, 4)
