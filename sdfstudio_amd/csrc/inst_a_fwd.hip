// BASELINE config 2/3 geometry network, forward + analytic-normal chain (three modes).
#include "field_inst.h"
SDFHIP_DEFINE_GEO_FWD(A, 8, 3, 6, 8, 4, 8)
