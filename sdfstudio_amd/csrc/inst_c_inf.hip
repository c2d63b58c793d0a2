#include "field_inst.h"

SDFHIP_DEFINE_GEO_FWD_INFER_HP(C, 8, 6, 8)  // + the 24-bit first-order forward (numerical normals at small delta)
