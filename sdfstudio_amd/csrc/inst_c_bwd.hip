#include "field_inst.h"

SDFHIP_DEFINE_GEO_BWD(C, 8, 6, 8)
