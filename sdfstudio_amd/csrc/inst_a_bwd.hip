#include "field_inst.h"

SDFHIP_DEFINE_GEO_BWD(A, 8, 3, 8)
