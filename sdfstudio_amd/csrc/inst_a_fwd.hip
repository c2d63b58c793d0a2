// BASELINE config 2/3 geometry network, forward + analytic-normal chain, training variant (saves z_l / r_l).
#include "field_inst.h"
SDFHIP_DEFINE_GEO_FWD_TRAIN(A, 8, 3, 8, 8, 4, 8)
