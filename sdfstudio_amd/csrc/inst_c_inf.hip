#include "field_inst.h"

SDFHIP_DEFINE_GEO_FWD_INFER(C, 8, 6, 8)
