// Explicit instantiations of the walker kernels, translation unit 16_2
// (lane-group shape G_P, Z = position-classified pairs); see qmc_inst.h.
#include "qmc_inst.h"
QMC_TU_16_2(QMC_NO_KW)
