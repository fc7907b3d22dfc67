// qmc_math.h -- fp64 elementary functions specialised for the walker engine.
//
// Every VALU instruction costs ~4 cycles per wave64 on gfx950 whatever its
// width (measured, profiles/r01_v1_dmc64_pmc_summary.txt), so the engine's
// speed is its instruction count.  The generic OCML routines carry argument
// reduction for the whole double range (Payne-Hanek), denormal/inf/nan paths
// and IEEE-exact division; the arguments here are bounded (positions in
// [0, L), angles of a few pi, |x| < 40 for exp) and finite, so these versions
// keep only what those ranges need.  Accuracy: <= 2 ulp, checked on the GPU
// against the CPU oracle through the parity tests (2e-11 relative on every
// energy / drift / log-psi).
#pragma once
#include "qmc_log_table.h"

#include <hip/hip_runtime.h>

// A polynomial coefficient pinned to a scalar register pair.  v_fma_f64 takes
// one SGPR operand, so a Horner step fma(p, z, C) is ONE vector instruction
// when C sits in SGPRs (two s_mov_b32 on the scalar unit).  Left alone the
// compiler, inside divergent or conditionally executed regions, materialises
// each constant with two v_mov_b32 to use the two-address v_fmac form: three
// vector instructions per step (measured: 18-24 extra per log / exp / sincos
// evaluation, tools/isa_one.sh).
__device__ __forceinline__ double sconst(double c)
{
    asm("" : "+s"(c));
    return c;
}

// q = x / y for normal, finite operands: hardware reciprocal estimate
// (v_rcp_f64, ~2^-25 relative error), one Newton step on the reciprocal, one
// correction of the quotient.  6 instructions against the 11 of the IEEE
// sequence (v_div_scale x2, v_rcp, 4 fma, mul, fma, v_div_fmas, v_div_fixup).
__device__ __forceinline__ double fast_div(double x, double y)
{
    double r = __builtin_amdgcn_rcp(y);
    double e = fma(-y, r, 1.0);
    r = fma(r, e, r);
    double q = x * r;
    double rem = fma(-y, q, x);
    return fma(rem, r, q);
}

// Pair-loop quotient: reciprocal estimate + ONE quotient correction (4
// instructions).  Relative error <= eps_rcp^2 + 2^-53 ~ 8e-16 (eps_rcp = 2^-25).
__device__ __forceinline__ double pair_div(double x, double y)
{
    double r = __builtin_amdgcn_rcp(y);
    double q = x * r;
    double rem = fma(-y, q, x);
    return fma(rem, r, q);
}

// float flavour of the pair quotient (reduced-precision pair loop): the
// hardware reciprocal is good to 1 ulp, which is what a float carries
__device__ __forceinline__ float pair_div(float x, float y)
{
    return x * __builtin_amdgcn_rcpf(y);
}

// type-generic helpers of the pair loop (double / float)
__device__ __forceinline__ double q_abs(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float q_abs(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ double q_fma(double a, double b, double c)
{
    return __builtin_fma(a, b, c);
}
__device__ __forceinline__ float q_fma(float a, float b, float c)
{
    return __builtin_fmaf(a, b, c);
}
__device__ __forceinline__ double q_copysign(double m, double s)
{
    return __builtin_copysign(m, s);
}
__device__ __forceinline__ float q_copysign(float m, float s)
{
    return __builtin_copysignf(m, s);
}
// split the binary exponent off a running product (no underflow, no overflow)
__device__ __forceinline__ void q_fold(double &p, int &e)
{
    e += __builtin_amdgcn_frexp_exp(p);
    p = __builtin_amdgcn_frexp_mant(p);
}
__device__ __forceinline__ void q_fold(float &p, int &e)
{
    e += __builtin_amdgcn_frexp_expf(p);
    p = __builtin_amdgcn_frexp_mantf(p);
}

__device__ __forceinline__ double fast_rcp(double y)
{
    double r = __builtin_amdgcn_rcp(y);
    double e = fma(-y, r, 1.0);
    r = fma(r, e, r);
    e = fma(-y, r, 1.0);
    return fma(r, e, r);
}

// sin and cos on [-pi/4, pi/4]: the classic degree-13 / degree-14 minimax
// kernels (coefficients of fdlibm's k_sin.c / k_cos.c, error < 2^-58).
__device__ __forceinline__ void sincos_kernel(double x, double &s, double &c)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double ps = fma(z, fma(z, fma(z, fma(z, fma(z, sconst(S6), sconst(S5)),
                                         sconst(S4)), sconst(S3)), sconst(S2)),
                    sconst(S1));
    s = fma(x * z, ps, x);
    double pc = fma(z, fma(z, fma(z, fma(z, fma(z, sconst(C6), sconst(C5)),
                                         sconst(C4)), sconst(C3)), sconst(C2)),
                    sconst(C1));
    c = fma(z * z, pc, fma(z, -0.5, 1.0));
}

// sin/cos of (pi/2) * u for a modest u (|u| < 2^20): quadrant split by
// round-to-nearest, kernels on the remainder.  u = 2 z / L gives the angle
// pi z / L with no cancellation; u = k2 z * (2/pi) gives k2 z.
__device__ __forceinline__ void sincos_halfpi(double u, double &s, double &c)
{
    double q = rint(u);
    double x = (u - q) * 1.57079632679489661923;   // |x| <= pi/4
    double ks, kc;
    sincos_kernel(x, ks, kc);
    int iq = (int)q;
    bool swap = iq & 1;
    double a = swap ? kc : ks;       // |sin|
    double b = swap ? ks : kc;       // |cos|
    // quadrant signs: sin negative for q = 2, 3; cos negative for q = 1, 2
    s = (iq & 2) ? -a : a;
    c = ((iq + 1) & 2) ? -b : b;
}

// exp(x) for |x| < 700 (no overflow/denormal handling): n = rint(x / ln 2),
// r = x - n ln2 (hi/lo split), degree-12 Taylor-minimax on |r| <= ln2/2.
__device__ __forceinline__ double exp_bounded(double x)
{
    const double INV_LN2 = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double n = rint(x * INV_LN2);
    double r = fma(-n, LN2_HI, x);
    r = fma(-n, LN2_LO, r);
    // exp(r) = sum r^k / k!, k <= 12: remainder (ln2/2)^13/13! ~ 1.7e-16
    double p = 2.08767569878680989792e-09;            // 1/12!
    p = fma(p, r, sconst(2.50521083854417187751e-08));        // 1/11!
    p = fma(p, r, sconst(2.75573192239858906526e-07));        // 1/10!
    p = fma(p, r, sconst(2.75573192239858906526e-06));        // 1/9!
    p = fma(p, r, sconst(2.48015873015873015873e-05));        // 1/8!
    p = fma(p, r, sconst(1.98412698412698412698e-04));        // 1/7!
    p = fma(p, r, sconst(1.38888888888888888889e-03));        // 1/6!
    p = fma(p, r, sconst(8.33333333333333333333e-03));        // 1/5!
    p = fma(p, r, sconst(4.16666666666666666667e-02));        // 1/4!
    p = fma(p, r, sconst(1.66666666666666666667e-01));        // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// log(x) for normal positive x: x = m 2^k with m in [1/2, 1) as
// v_frexp_mant_f64 / v_frexp_exp_i32_f64 deliver them; log m = L_r + log1p(d),
// d = m inv_c_r - 1, from a 256-row table of {inv_c, L = -log(inv_c)} (row
// centre c, inv_c the double nearest 1/c, L computed from that double: the
// identity is exact) and five terms of log1p (|d| <= 1.95e-3: the sixth is
// 1e-17).  16 instructions, no division, against 30 + v_rcp_f64 for the atanh
// series this replaced.  Just above x = 1 (m just above 1/2, k = 1) the result
// is the difference of k ln 2 and log m: its ABSOLUTE error stays at the
// 1e-16 of those terms, which is what the callers need -- they sum logarithms
// of pair products into log|psi| ~ 10^2..10^3 -- and what the bound states:
// worst deviation from long-double log 1.25e-16 (1 + |log x|)
// (qmc_log_table_info, tests/test_cabi.py).  (Round 3 moved the mantissa to
// [sqrt(1/2), sqrt(2)) to keep the RELATIVE error near 1 as well: four
// instructions per logarithm for a property nobody uses.)
#ifndef QMC_LOG_TABLE
#define QMC_LOG_TABLE 1
#endif
static __device__ const double QMC_LOG_TAB[2 * QMC_LOG_ROWS] = { QMC_LOG_TAB_VALUES };

__device__ __forceinline__ double log_series(double x)
{
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    int e;
    double m = frexp(x, &e);                 // m in [0.5, 1)
    if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }
    double s = fast_div(m - 1.0, m + 1.0);
    double z = s * s;
    double p = sconst(1.0 / 21.0);
    p = fma(p, z, sconst(1.0 / 19.0));
    p = fma(p, z, sconst(1.0 / 17.0));
    p = fma(p, z, sconst(1.0 / 15.0));
    p = fma(p, z, sconst(1.0 / 13.0));
    p = fma(p, z, sconst(1.0 / 11.0));
    p = fma(p, z, sconst(1.0 / 9.0));
    p = fma(p, z, sconst(1.0 / 7.0));
    p = fma(p, z, sconst(1.0 / 5.0));
    p = fma(p, z, sconst(1.0 / 3.0));
    double lm = fma(s * z, 2.0 * p, 2.0 * s);
    double de = (double)e;
    return fma(de, LN2_HI, fma(de, LN2_LO, lm));
}

__device__ __forceinline__ double log_pos(double x)
{
    if (!QMC_LOG_TABLE) return log_series(x);
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    const double m = __builtin_amdgcn_frexp_mant(x);       // [1/2, 1)
    const int k = __builtin_amdgcn_frexp_exp(x);
    // row: truncation of (m - 1/2) 2 ROWS, in [0, ROWS) for every m
    // (masked: an argument that is not a positive normal number -- never one
    // whose result is used -- still reads inside the table)
    const int r = (int)fma(m, (double)(2 * QMC_LOG_ROWS), -(double)QMC_LOG_ROWS) &
                  (QMC_LOG_ROWS - 1);
    typedef const __attribute__((address_space(1))) double *gptr;
    const gptr row = (gptr)QMC_LOG_TAB + 2u * (unsigned)r;
    const double inv_c = row[0], L = row[1];
    const double d = fma(m, inv_c, -1.0);
    double q = fma(d, sconst(0.2), -0.25);
    q = fma(q, d, sconst(1.0 / 3.0));
    q = fma(q, d, -0.5);
    const double lm = fma(d * d, q, d) + L;
    const double kd = (double)k;
    return fma(kd, LN2_HI, fma(kd, LN2_LO, lm));
}

__device__ __forceinline__ double fast_sqrt(double x)
{
    // v_rsq_f64 estimate + two Newton steps (x > 0, normal)
    double r = __builtin_amdgcn_rsq(x);
    double g = x * r;
    double h = 0.5 * r;
    double d = fma(-h, g, 0.5);
    g = fma(g, d, g);
    h = fma(h, d, h);
    d = fma(-g, g, x);
    return fma(d, h, g);
}
