// Explicit instantiations of the walker kernels, translation unit 64_4_Z0
// (lane-group shape G_P, Z = position-classified pairs); see qmc_inst.h.
#include "qmc_inst.h"
QMC_TU_64_4_Z0(QMC_NO_KW)
