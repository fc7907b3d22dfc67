// qmc_inst.h -- the list of kernel instantiations, grouped into translation
// units (inst_*.hip: explicit instantiation definitions, compiled in parallel
// by `make -j`); qmcwalk.hip sees them as `extern template` declarations.
//
// Which (PAD, ZC) variants exist follows the dispatch in qmcwalk.hip: shapes
// with P >= 4 always run the masked (PAD) variant (LaunchX::want_mask).
#pragma once

#include "qmc_kernels.h"

#define QMC_INST_EPV_R(KW, G, P, PAD, ZC, R)                                  \
    KW template __global__ void evaluate_kernel<G, P, PAD, ZC, R>(            \
        const DevModel *, EvalArgs);                                          \
    KW template __global__ void prepare_kernel<G, P, PAD, ZC, R>(             \
        const DevModel *, PrepArgs);                                          \
    KW template __global__ void vmc_step_kernel<G, P, PAD, ZC, true, R>(      \
        const DevModel *, VmcArgs);                                           \
    KW template __global__ void vmc_step_kernel<G, P, PAD, ZC, false, R>(     \
        const DevModel *, VmcArgs);
#define QMC_INST_EVO_R(KW, G, P, PAD, ZC, R)                                  \
    KW template __global__ void dmc_evolve_kernel<G, P, PAD, ZC, R>(          \
        const DevModel *, EvolveArgs);
#define QMC_INST_EPV(KW, G, P, PAD, ZC) QMC_INST_EPV_R(KW, G, P, PAD, ZC, double)
#define QMC_INST_EVO(KW, G, P, PAD, ZC) QMC_INST_EVO_R(KW, G, P, PAD, ZC, double)
#define QMC_INST_ALL(KW, G, P, PAD, ZC)                                       \
    QMC_INST_EPV(KW, G, P, PAD, ZC) QMC_INST_EVO(KW, G, P, PAD, ZC)
// the reduced-precision (float pair loop) variants: one wavefront per walker
// shapes, pairs classified from the sines (ZC = false)
#define QMC_INST_ALL_F(KW, G, P, PAD)                                         \
    QMC_INST_EPV_R(KW, G, P, PAD, false, float)                               \
    QMC_INST_EVO_R(KW, G, P, PAD, false, float)

// small shapes: every variant
#define QMC_INST_SMALL(KW, G, P)                                              \
    QMC_INST_ALL(KW, G, P, false, false) QMC_INST_ALL(KW, G, P, false, true)  \
    QMC_INST_ALL(KW, G, P, true, false) QMC_INST_ALL(KW, G, P, true, true)

#define QMC_TU_16_1(KW) QMC_INST_SMALL(KW, 16, 1)
#define QMC_TU_16_2(KW) QMC_INST_SMALL(KW, 16, 2)
#define QMC_TU_32_2(KW) QMC_INST_SMALL(KW, 32, 2)
#define QMC_TU_64_1(KW) QMC_INST_SMALL(KW, 64, 1)
#define QMC_TU_64_2(KW) QMC_INST_SMALL(KW, 64, 2)
#define QMC_TU_64_4_Z0(KW) QMC_INST_ALL(KW, 64, 4, true, false)
#define QMC_TU_64_4_Z1(KW) QMC_INST_ALL(KW, 64, 4, true, true)
#define QMC_TU_64_8_Z0(KW) QMC_INST_ALL(KW, 64, 8, true, false)
#define QMC_TU_64_8_Z1(KW) QMC_INST_ALL(KW, 64, 8, true, true)

#define QMC_TU_F32_64_1(KW)                                                   \
    QMC_INST_ALL_F(KW, 64, 1, false) QMC_INST_ALL_F(KW, 64, 1, true)
#define QMC_TU_F32_64_2(KW)                                                   \
    QMC_INST_ALL_F(KW, 64, 2, false) QMC_INST_ALL_F(KW, 64, 2, true)
#define QMC_TU_F32_64_4(KW) QMC_INST_ALL_F(KW, 64, 4, true)
#define QMC_TU_F32_64_8(KW) QMC_INST_ALL_F(KW, 64, 8, true)

#define QMC_NO_KW
#define QMC_FOR_ALL_TUS(X)                                                    \
    X(16_1) X(16_2) X(32_2) X(64_1) X(64_2) X(64_4_Z0) X(64_4_Z1)            \
    X(64_8_Z0) X(64_8_Z1) X(F32_64_1) X(F32_64_2) X(F32_64_4) X(F32_64_8)
// every shape pick_shape() can return
#define QMC_FOR_ALL_SHAPES(X)                                                 \
    X(16, 1) X(16, 2) X(32, 2) X(64, 1) X(64, 2) X(64, 4) X(64, 8)
