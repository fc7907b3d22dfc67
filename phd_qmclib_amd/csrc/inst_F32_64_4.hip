// Explicit instantiations of the walker kernels, translation unit F32_64_4
// (reduced-precision pair loop, lane-group shape G_P); see qmc_inst.h.
#include "qmc_inst.h"
QMC_TU_F32_64_4(QMC_NO_KW)
