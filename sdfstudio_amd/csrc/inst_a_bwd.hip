// BASELINE config 2/3 geometry network, tangent pass + data backward.
#include "field_inst.h"
SDFHIP_DEFINE_GEO_BWD(A, 8, 3, 8, 8, 4, 8)
