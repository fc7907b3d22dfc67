// Explicit instantiations of the walker kernels, translation unit 64_1
// (lane-group shape G_P, Z = position-classified pairs); see qmc_inst.h.
#include "qmc_inst.h"
QMC_TU_64_1(QMC_NO_KW)
