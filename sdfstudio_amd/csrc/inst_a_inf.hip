#include "field_inst.h"

SDFHIP_DEFINE_GEO_FWD_INFER(A, 8, 3, 8)
