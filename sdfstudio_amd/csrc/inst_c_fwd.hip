#include "field_inst.h"

SDFHIP_DEFINE_GEO_FWD_TRAIN(C, 8, 6, 8)
