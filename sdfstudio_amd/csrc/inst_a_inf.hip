// BASELINE config 2/3 geometry network, inference variants (geonetwork: sdf + feature; sdf only).
#include "field_inst.h"
SDFHIP_DEFINE_GEO_FWD_INFER(A, 8, 3, 8, 8, 4, 8)
