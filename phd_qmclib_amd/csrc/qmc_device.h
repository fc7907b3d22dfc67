// qmc_device.h -- device-side building blocks of the gfx950 walker engine.
//
// Work mapping (CDNA4, wave64): one walker is owned by a GROUP of G lanes
// (G = 64: one wavefront per walker; G = 16: four walkers per wavefront) and
// every lane keeps P particles in registers (N <= G*P).  The O(N^2) Jastrow
// pair sum runs as a systolic rotation over lane offsets k = 1..G/2: at step
// k lane l pairs its P particles with the P particles of lane (l-k) mod G,
// whose per-particle sin/cos tables it reads from LDS; the contribution to the
// partner's drift travels in a register that rotates one lane per step, so
// every unordered pair is evaluated exactly once (N(N-1)/2 pair evaluations
// instead of the reference's N(N-1), qmc_base/jastrow/model.py:834-848).
//
// Pair arithmetic: no transcendental per pair.  With a_i = pi z_i / L and
// b_i = k2 z_i tabulated per particle (sin, cos), the angle-difference
// identities give sin/cos of a_i-a_j and b_i-b_j with 2 FMA-class ops each.
// Both branches of the two-body factor (mrbp_qmc/model.py:468-529) reduce to
// ONE division q = X / Y per pair with the drift coefficient folded into X:
//   long  (r >= rm): X = (pi/L) beta cos(a_i-a_j), Y = sin(a_i-a_j)
//                    (period L: the minimum image needs no explicit wrap)
//   short (r <  rm): X = -k2 sin(k2 d -+ phi),      Y = cos(k2 r - phi)
// so that q is the pair's contribution to the drift of particle i, and
//   -f2''/f2 + (f2'/f2)^2 = k2^2 + q^2                   (short)
//                         = (pi/L)^2 beta + q^2 / beta   (long).
// The short branch runs under an exec mask (a real divergent branch, no
// selects).
//
// On gfx950 every VALU instruction costs ~4 cycles per wave64 and this path is
// VALU-bound (profiles/): instruction count is the whole game here.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qmc_math.h"

#define QMC_PI 3.141592653589793238462643383279502884

// Section markers for the static instruction census (tools/isa_one.sh builds
// with -DQMC_SECTIONS; the production build emits nothing).
#ifdef QMC_SECTIONS
#define QMC_SECTION(name) asm volatile("; SECTION " name)
#else
#define QMC_SECTION(name) do { } while (0)
#endif

struct DevModel {
    int n;                 // boson_number
    int is_free, is_ideal;
    int defects_sep;
    int zclass;            // classify pairs from positions (rm close to L/2)
    double L, half_L, rm, L_minus_rm;
    double two_over_L;     // 2 / L            (angle pi z / L = (pi/2) * u)
    double k2_2pi;         // k2 * 2 / pi      (angle k2 z     = (pi/2) * u)
    double k2, k2sq;
    double m_k2cphi;       // -k2 cos(k2 r_off)
    double k2sphi;         //  k2 sin(k2 r_off)
    double cphi, sphi;     // cos/sin(k2 r_off)
    double cth, sth;       // cos(k2 L), |sin(k2 L)|
    int sth_sign;          // sign bit of sin(k2 L) (0 or 0x80000000)
    // short-range variants (pair_core4): angles c_v added to k2 z_own,
    // v = (no wrap, d > 0), (no wrap, d < 0), (D > L/2), (D < -L/2)
    double var_cos[4], var_sin[4];
    double m_k2;           // -k2
    double sin_rm;         // sin(pi rm / L)
    double a_long, b_long; // (pi/L) beta, (pi/L)^2 beta
    double inv_beta;       // 1 / beta
    double beta, log_am;
    // one-body (Kronig-Penney)
    double z_a, z_b, k1, kp1, e0, v0, v0d, v0_minus_e0, cf;
    int uniform_barrier;   // every barrier has the same height v_barrier
    double v_barrier;
    double k1_2pi;         // k1 * 2 / pi
    double k1_half;        // k1 / 2
};

// ---------------------------------------------------------------- RNG ----
// Philox4x32-10 (Salmon et al. 2011), counter = (slot, step, index, stream),
// key = seed.  Same algorithm and keying as the oracle so seeded runs line up.
enum { STREAM_VMC_MOVE = 0, STREAM_VMC_ACCEPT = 1, STREAM_DMC_BRANCH = 2,
       STREAM_DMC_DIFFUSE = 3 };

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0,
                                              uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply each (v_mad_u64_u32)
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo)
{
    uint64_t b = ((uint64_t)(hi >> 5) << 26) | (uint64_t)(lo >> 6);
    return (double)b * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ void philox_uniform2(uint64_t seed, uint32_t slot,
                                                uint32_t step, uint32_t index,
                                                uint32_t stream, double &u0,
                                                double &u1)
{
    uint32_t c[4] = { slot, step, index, stream };
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    u0 = u53(c[0], c[1]);
    u1 = u53(c[2], c[3]);
}

// Box-Muller pair from one Philox block: both standard normals.
__device__ __forceinline__ void philox_normal2(uint64_t seed, uint32_t slot,
                                               uint32_t step, uint32_t index,
                                               uint32_t stream, double &g0,
                                               double &g1)
{
    double u0, u1;
    philox_uniform2(seed, slot, step, index, stream, u0, u1);
    // 1 - u0 is in (0, 1]: log <= 0
    double r = fast_sqrt(fmax(-2.0 * log_pos(1.0 - u0), 1e-300));
    double s, c;
    sincos_halfpi(4.0 * u1, s, c);          // angle 2 pi u1
    g0 = r * c;
    g1 = r * s;
}

// ------------------------------------------------------------ helpers ----
// Periodic wrap into [0, L) with the reference's floor-mod result
// (qmc_base/utils.py:55-66) for excursions of less than one box length.
__device__ __forceinline__ double wrap_box(double z, double L)
{
    if (z < 0.0) {
        z = (z >= -L) ? z + L : z - L * floor(z / L);
    } else if (z >= L) {
        z = (z < 2.0 * L) ? z - L : z - L * floor(z / L);
    }
    return z;
}

template <int G>
__device__ __forceinline__ double group_sum(double v)
{
#pragma unroll
    for (int m = 1; m < G; m <<= 1)
        v += __shfl_xor(v, m, 64);
    return v;
}

// Lane order = position order.  Bosons are identical, so which lane holds which
// particle is free; when the lanes of a group hold the particles in (cyclic)
// position order, the lanes met at rotation step k all sit at about the same
// separation k L / N and the short-range branch of the pair loop becomes
// (nearly) wave-uniform: whole steps skip it through s_cbranch_execz (+10 %
// measured).  A label per lane remembers the particle's original index: RNG
// counters, tapes and every array handed back to the host use the label, so
// results do not depend on the lane order (only summation order does).
// One odd-even transposition pass per time step keeps the order as particles
// diffuse; the comparison is on the minimum-image separation, so a particle
// that crosses the box boundary stays correctly (cyclically) ordered.
template <int G, int P>
__device__ __forceinline__ void resort_step(double (&z)[P], int (&lab)[P],
                                            int gl, unsigned parity, int n,
                                            double L, double half_L)
{
    const int lane = threadIdx.x & 63, base = lane - gl;
    // partner in the row: even phase (0,1)(2,3)..; odd phase (1,2)(3,4)..
    // and, across the row seam, (G-1 of row a, 0 of row a+1) cyclically
    int pg = (parity & 1u) ? ((gl & 1) ? gl + 1 : gl - 1) : (gl ^ 1);
    const bool wrap_hi = pg >= G, wrap_lo = pg < 0;
    if (wrap_hi) pg = 0;
    if (wrap_lo) pg = G - 1;
    double z0[P]; int l0[P];
#pragma unroll
    for (int a = 0; a < P; ++a) { z0[a] = z[a]; l0[a] = lab[a]; }
#pragma unroll
    for (int a = 0; a < P; ++a) {
        const int want = wrap_hi ? (a + 1) % P : (wrap_lo ? (a + P - 1) % P : a);
        double zp = 0.0; int lp = 0;
#pragma unroll
        for (int b = 0; b < P; ++b) {
            // (shuffles are executed by every lane; each keeps the row it needs)
            double zz = __shfl(z0[b], base + pg, 64);
            int ll = __shfl(l0[b], base + pg, 64);
            if (b == want) { zp = zz; lp = ll; }
        }
        const bool valid = (gl + G * a) < n && (pg + G * want) < n;
        // "lower" = the element whose rank comes first in the cyclic order
        const bool lower = wrap_hi ? true : (wrap_lo ? false : gl < pg);
        double d = lower ? zp - z0[a] : z0[a] - zp;   // upper minus lower
        if (d > half_L) d -= L;
        if (d < -half_L) d += L;
        if (valid && d < 0.0) { z[a] = zp; lab[a] = lp; }
    }
}

// Per-particle table entry kept in registers by the owner and published to LDS.
struct PTab {
    double s, c;    // sin/cos(pi z / L)
    double su, cu;  // sin/cos(k2 z)
};

// One-body factor (mrbp_qmc/model.py:404-464) and lattice potential (:533-551).
// ldz = f1'/f1; kin_pot = -f1''/f1 + ldz^2 + V(z); the factor itself is
// f1 * exp(-xoff) > 0 (the barrier's cosh x is returned as (e^{2x} + 1) / 2
// with xoff = x: the caller needs log f1 only and subtracts xoff there).
__device__ __forceinline__ void one_body(const DevModel &m, double z,
                                         double &ldz, double &kin_pot,
                                         double &f1, double &xoff)
{
    double n_cell = floor(z);
    double z_cell = z - n_cell;
    if (m.z_a < z_cell) {
        // barrier: tanh x = (e^{2x} - 1) / (e^{2x} + 1), one exponential and
        // one division; cosh x = (e^{2x} + 1) / 2 * e^{-x}
        double x = m.kp1 * (z_cell - 1.0 + 0.5 * m.z_b);
        double e2 = exp_bounded(2.0 * x);
        double den = e2 + 1.0;
        ldz = m.kp1 * fast_div(e2 - 1.0, den);
        f1 = 0.5 * den;
        xoff = x;
        double v = m.v_barrier;
        if (!m.uniform_barrier) {
            // lattice defects: every defects_sep-th barrier has height v0d
            // (an integer modulo by a run-time divisor is ~25 instructions,
            // skipped by the whole wave in the common defect-free case)
            int nc = (int)n_cell;
            int r = nc % m.defects_sep;
            if (r < 0) r += m.defects_sep;
            v = (r == 0) ? m.v0d : m.v0;
        }
        kin_pot = fma(ldz, ldz, v - m.v0_minus_e0);
    } else {
        // |k1 (z_cell - z_a/2)| < pi/2 (the ground band's cosine has no node
        // in the well), so half the angle fits the sin/cos kernels without
        // range reduction or quadrant logic:
        //   sin x = 2 s c,  cos x = (c - s)(c + s),  s, c = sin, cos(x / 2)
        double sh, ch;
        sincos_kernel(m.k1_half * (z_cell - 0.5 * m.z_a), sh, ch);
        const double sx = 2.0 * sh * ch;
        const double cx = (ch - sh) * (ch + sh);
        ldz = -m.k1 * fast_div(sx, cx);
        f1 = m.cf * cx;
        xoff = 0.0;
        kin_pot = fma(ldz, ldz, m.e0);
    }
}

// Constants of the pair loop, loaded once per walker evaluation.  The two that
// feed a copysign live in VGPRs (a v_bfi on the high dword then needs no move
// of the low dword from an SGPR every pair).
struct PairConsts {
    double sin_rm, cth, m_k2cphi, sphi, cphi;
    double v_sth, v_k2sphi;       // VGPR-resident
    double half_L, rm, L_minus_rm;
    double m_k2;
    int sth_sign;
};

__device__ __forceinline__ PairConsts load_pair_consts(const DevModel &m)
{
    PairConsts c;
    c.sin_rm = m.sin_rm; c.cth = m.cth; c.m_k2cphi = m.m_k2cphi;
    c.sphi = m.sphi; c.cphi = m.cphi;
    c.half_L = m.half_L; c.rm = m.rm; c.L_minus_rm = m.L_minus_rm;
    c.v_sth = m.sth; c.v_k2sphi = m.k2sphi;
    c.sth_sign = m.sth_sign;
    c.m_k2 = m.m_k2;
    asm volatile("" : "+v"(c.v_sth), "+v"(c.v_k2sphi));
    return c;
}

// One pair, seen from the own particle (table `a`, long-range numerator
// coefficients aks/akc = a_long * (sin, cos)) against partner table `b`.
//   q       : contribution to the drift of the own particle (partner: -q)
//   Yout    : the denominator = the factor |f2| up to constants
//             (short: cos(k2 r - phi); long: sin(pi d / L), signed)
//   isshort : r < rm
template <bool ZCLASS>
__device__ __forceinline__ void pair_core(const PairConsts &m, const PTab &a,
                                          double aks, double akc, double za,
                                          const PTab &b, double zb, double &q,
                                          double &Yout, bool &isshort,
                                          unsigned long long &shortmask)
{
    double S = a.s * b.c - a.c * b.s;     // sin(pi (z_a - z_b) / L)
    double X = akc * b.c + aks * b.s;     // a_long * cos(...)
    double Y = S;
    bool wrapped;
    if (ZCLASS) {
        double aD = fabs(za - zb);
        wrapped = aD > m.half_L;
        isshort = (aD < m.rm) | (aD > m.L_minus_rm);
    } else {
        wrapped = X < 0.0;                // |z_a - z_b| > L/2
        isshort = fabs(S) < m.sin_rm;     // min-image r < rm
    }
    // taken here, in the block of the compare, the ballot is the compare's own
    // SGPR mask (later it costs a v_cndmask + v_cmp round trip)
    shortmask = __ballot(isshort);
    if (isshort) {
        double Su = a.su * b.cu - a.cu * b.su;   // sin(k2 (z_a - z_b))
        double Cu = a.cu * b.cu + a.su * b.su;
        if (wrapped) {
            // keep this a real (exec-masked) branch: as selects it costs four
            // v_cndmask on top of the arithmetic
            asm volatile("");
            // min image d = D - sgn(D) L; sgn(D) = sgn(S)
            // t = sin(k2 L) sgn(S): copysign on |sin(k2 L)|, then the sign of
            // sin(k2 L) itself (k2 L is any angle) xor-ed into the high word
            double t = __builtin_copysign(m.v_sth, S);
            t = __hiloint2double(__double2hiint(t) ^ m.sth_sign,
                                 __double2loint(t));
            double ct, st;
            const double cth = m.cth;
            // in place, exactly four instructions (the compiler's two-address
            // v_fmac form needs two extra 64-bit moves at the join)
            asm("v_mul_f64 %[ct], %[cu], %[t]\n\t"
                "v_mul_f64 %[st], %[su], %[t]\n\t"
                "v_fma_f64 %[su], %[su], %[cth], -%[ct]\n\t"
                "v_fma_f64 %[cu], %[cu], %[cth], %[st]"
                : [su] "+v"(Su), [cu] "+v"(Cu), [ct] "=&v"(ct), [st] "=&v"(st)
                : [t] "v"(t), [cth] "s"(cth));
        }
        // now (Su, Cu) = sin/cos(k2 d), |k2 d| < pi/2, sgn(Su) = sgn(d):
        //   -k2 tan(k2 r - phi) sgn(d) = X / Y with
        double t2 = __builtin_copysign(m.v_k2sphi, Su);
        X = fma(m.m_k2cphi, Su, Cu * t2);
        Y = fma(fabs(Su), m.sphi, Cu * m.cphi);
    }
    q = pair_div(X, Y);
    Yout = Y;
}

// Short-range pair, four-case form (used for P <= 2).  With the minimum-image
// separation d = D - w L (w = 0, +1, -1) and s = sgn(d), the pair needs
//   X = -k2 sin(theta), Y = cos(theta), theta = k2 d - phi s
//     = (k2 z_own - k2 w L - phi s) - k2 z_partner = A_own(w, s) - k2 z_partner
// and only four (w, s) combinations exist: D in (0, L/2] -> (0, +),
// [-L/2, 0) -> (0, -), (L/2, L) -> (+1, -), (-L, -L/2) -> (-1, +).  The own
// particle carries sin/cos of its four angles A (16 instructions per particle
// and step); a pair is then 4 instructions in the branch of its case instead
// of the rotate-by-k2 L / copysign sequence.  With position-sorted lanes
// nearly every lane of a rotation step is in the same one or two cases.
struct ShortTab {
    double s[4], c[4];
};

__device__ __forceinline__ void make_short_tab(const DevModel &m, double su,
                                               double cu, ShortTab &st)
{
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        st.s[v] = fma(su, m.var_cos[v], cu * m.var_sin[v]);
        st.c[v] = fma(cu, m.var_cos[v], -(su * m.var_sin[v]));
    }
}

template <bool ZCLASS>
__device__ __forceinline__ void pair_core4(const PairConsts &m, const PTab &a,
                                           const ShortTab &sa, double aks,
                                           double akc, double za,
                                           const PTab &b, double zb, double &q,
                                           double &Yout, bool &isshort,
                                           unsigned long long &shortmask)
{
    double S = a.s * b.c - a.c * b.s;     // sin(pi (z_a - z_b) / L)
    double X = akc * b.c + aks * b.s;     // a_long * cos(...)
    double Y = S;
    bool wrapped, neg;
    if (ZCLASS) {
        const double D = za - zb;
        const double aD = fabs(D);
        wrapped = aD > m.half_L;
        neg = D < 0.0;
        isshort = (aD < m.rm) | (aD > m.L_minus_rm);
    } else {
        wrapped = X < 0.0;                // |z_a - z_b| > L/2
        neg = S < 0.0;                    // sgn(D) = sgn(sin(pi D / L))
        isshort = fabs(S) < m.sin_rm;     // min-image r < rm
    }
    shortmask = __ballot(isshort);
    if (isshort) {
        double xs, ys;
        // real (exec-masked) branches: as selects the four cases would cost
        // eight v_cndmask per double pair
#define QMC_CASE4(v)                                                          \
        {                                                                     \
            asm volatile("");                                                 \
            xs = sa.s[v] * b.cu - sa.c[v] * b.su;                             \
            ys = sa.c[v] * b.cu + sa.s[v] * b.su;                             \
        }
        if (!wrapped) {
            if (!neg) QMC_CASE4(0) else QMC_CASE4(1)
        } else {
            if (!neg) QMC_CASE4(2) else QMC_CASE4(3)
        }
#undef QMC_CASE4
        X = m.m_k2 * xs;
        Y = ys;
    }
    q = pair_div(X, Y);
    Yout = Y;
}

// LDS table of one lane group: 4 (5 with ZCLASS: + positions) arrays of
// DUP*G*P doubles.  For P = 1 every entry is stored twice (lane g at g and
// G + g) so a rotated read (g - k) never needs a modulo; for P >= 2 the copy
// costs occupancy through LDS (P = 8: 128 KB per block, one wave per SIMD), so
// the table is stored once and the rotated index is masked (one v_and per
// partner table, i.e. per 2-8 pairs).
template <int G, int P, bool ZCLASS>
struct GroupLds {
    static constexpr int DUP = (P >= 2) ? 1 : 2;
    static constexpr int ROW = DUP * G * P;
    static constexpr int DOUBLES = (ZCLASS ? 5 : 4) * ROW;
};

// Evaluate one walker held in registers.
//   z[P]      : positions owned by this lane (particle index gl + G*a)
//   F[P]      : out, drift of the own particles
//   eith[P]   : out if ITH, local energy per particle
//   E         : out, local energy of the walker (same value in every lane)
//   logwf     : out if WF, log|psi| (same value in every lane)
template <int G, int P, bool PAD, bool WF, bool ITH, bool ZCLASS>
__device__ __forceinline__ void eval_walker(const DevModel &m,
                                            const double (&z)[P], int gl,
                                            double *lds, double (&F)[P],
                                            double (&eith)[P], double &E,
                                            double &logwf)
{
    constexpr int DUP = GroupLds<G, P, ZCLASS>::DUP;
    constexpr int ROW = GroupLds<G, P, ZCLASS>::ROW;
    // Own particles are processed PA at a time: with P = 8 the tables of all
    // eight (96 VGPRs) would leave one wave per SIMD, so the rotation runs in
    // two passes of four own particles (tables re-read from LDS).
    constexpr int PA = (P > 4) ? 4 : P;
    constexpr int NPASS = P / PA;
    double *lS = lds, *lC = lds + ROW, *lSU = lds + 2 * ROW,
           *lCU = lds + 3 * ROW, *lZ = lds + 4 * ROW;
    const int n = m.n;
    // four-case short-range form while the own tables fit (see pair_core4)
    constexpr bool FOURCASE = (P <= 2);
    PTab t[PA];
    ShortTab st4[FOURCASE ? PA : 1];
    double aks[PA], akc[PA];   // a_long * (sin, cos)(pi z / L)
    bool ok[P];
    double kin1[P];          // one-body kinetic + potential (ITH)
    double kin1_sum = 0.0;   // their sum over the own particles (!ITH)
    double prodS = 1.0, prodL = 1.0;   // running products of pair factors (WF)
    double prod1 = 1.0;      // product of the one-body factors (WF)
    double xoff_sum = 0.0;   // sum of the exponents split off them (one_body)
    int expS = 0, expL = 0;  // binary exponents split off the products
    int exp1 = 0;            // ... and off the one-body product (P >= 4)
    int nshort = 0, npair = 0;
    // one walker per wavefront and no padding: short pairs are counted with a
    // ballot + scalar popcount (SALU) instead of a per-lane VALU add
    constexpr bool WAVE_COUNT = (G == 64) && !PAD;
    int ns_wave = 0;
    double Qall = 0.0, Qs = 0.0;  // sum of q^2 over all / short pairs
    double Kown[P], KT[P];   // per-particle pair kinetic sums (ITH)
    double T[P];             // travelling drift of the partner lane

    QMC_SECTION("tables+onebody");
#pragma unroll
    for (int a = 0; a < P; ++a) {
        ok[a] = !PAD || (gl + G * a) < n;
        F[a] = 0.0; T[a] = 0.0; Kown[a] = 0.0; KT[a] = 0.0;
        if (ITH) kin1[a] = 0.0;
        if (!m.is_ideal) {
            PTab ta;
            sincos_halfpi(z[a] * m.two_over_L, ta.s, ta.c);
            sincos_halfpi(z[a] * m.k2_2pi, ta.su, ta.cu);
            if (NPASS == 1) {
                t[a % PA] = ta;
                aks[a % PA] = m.a_long * ta.s;
                akc[a % PA] = m.a_long * ta.c;
                if (FOURCASE) make_short_tab(m, ta.su, ta.cu, st4[a % PA]);
            }
            int i0 = a * DUP * G + gl;
            lS[i0] = ta.s; lC[i0] = ta.c; lSU[i0] = ta.su; lCU[i0] = ta.cu;
            if (ZCLASS) lZ[i0] = z[a];
            if (DUP == 2) {
                lS[i0 + G] = ta.s; lC[i0 + G] = ta.c;
                lSU[i0 + G] = ta.su; lCU[i0 + G] = ta.cu;
                if (ZCLASS) lZ[i0 + G] = z[a];
            }
        } else if (NPASS == 1) {
            aks[a % PA] = 0.0; akc[a % PA] = 0.0;
        }
        if (!m.is_free) {
            double ldz, kp, f1, xoff;
            one_body(m, z[a], ldz, kp, f1, xoff);
            if (ok[a]) {
                F[a] = ldz;
                if (ITH) kin1[a] = kp; else kin1_sum += kp;
                if (WF) {
                    prod1 *= f1;
                    xoff_sum += xoff;
                    if (P >= 4) {     // many factors up to e^{2x} / 2 each
                        exp1 += __builtin_amdgcn_frexp_exp(prod1);
                        prod1 = __builtin_amdgcn_frexp_mant(prod1);
                    }
                }
            }
        }
    }

    // Split the binary exponent off a running product so that it can neither
    // underflow nor overflow (one frexp pair instead of a log per fold).
#define QMC_FOLD(p, e)                                                        \
    do {                                                                      \
        e += __builtin_amdgcn_frexp_exp(p);                                   \
        p = __builtin_amdgcn_frexp_mant(p);                                   \
    } while (0)
    // own table of particle (lane gl, register a) back from LDS
#define QMC_LOAD_OWN(dst, a)                                                  \
    do {                                                                      \
        const int i0_ = (a) * DUP * G + gl;                                   \
        (dst).s = lS[i0_]; (dst).c = lC[i0_];                                 \
        (dst).su = lSU[i0_]; (dst).cu = lCU[i0_];                             \
    } while (0)

    if (!m.is_ideal) {
        const PairConsts pc = load_pair_consts(m);
        // make the table visible to the other lanes of the wave (one wave owns
        // its groups' LDS region: LDS ops of a wave complete in order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // sums that count every unordered pair once
#define QMC_TALLY(q, Y, isshort)                                              \
    do {                                                                      \
        Qall = fma(q, q, Qall);                                               \
        if (!WAVE_COUNT) ++npair;                                             \
        /* prodL runs over ALL pairs (no else branch, no second compare);   \
           the short factors are divided out once at the end */             \
        if (WF) prodL *= Y;          /* sign dropped at the end */            \
        if (isshort) {                                                        \
            asm volatile("");   /* exec-masked, not selects */                \
            Qs = fma(q, q, Qs);                                               \
            if (!WAVE_COUNT) ++nshort;                                        \
            if (WF) prodS *= Y;                                               \
        }                                                                     \
    } while (0)
#define QMC_PAIR_KIN(q, isshort)                                              \
    ((isshort) ? fma(q, q, m.k2sq) : fma((q) * (q), m.inv_beta, m.b_long))

        // ---- k = 0: pairs inside the lane ----
        QMC_SECTION("pairs_in_lane");
#pragma unroll
        for (int a = 0; a < P; ++a) {
            PTab ta; double aksa, akca;
            if (NPASS == 1) {
                ta = t[a % PA]; aksa = aks[a % PA]; akca = akc[a % PA];
            } else if (a + 1 < P) {
                QMC_LOAD_OWN(ta, a);
                aksa = m.a_long * ta.s; akca = m.a_long * ta.c;
            }
#pragma unroll
            for (int b = a + 1; b < P; ++b) {
                PTab tb;
                if (NPASS == 1) tb = t[b % PA];
                else QMC_LOAD_OWN(tb, b);
                double q, Y; bool sh; unsigned long long shm;
                if (FOURCASE)
                    pair_core4<ZCLASS>(pc, ta, st4[a % PA], aksa, akca, z[a],
                                       tb, z[b], q, Y, sh, shm);
                else
                    pair_core<ZCLASS>(pc, ta, aksa, akca, z[a], tb, z[b], q,
                                      Y, sh, shm);
                if (WAVE_COUNT)
                    ns_wave += __popcll(shm);
                if (!PAD || (ok[a] && ok[b])) {
                    F[a] += q; F[b] -= q;
                    QMC_TALLY(q, Y, sh);
                    if (ITH) {
                        double kk = QMC_PAIR_KIN(q, sh);
                        Kown[a] += kk; Kown[b] += kk;
                    }
                }
            }
            if (WF && P > 4) { QMC_FOLD(prodS, expS); QMC_FOLD(prodL, expL); }
        }

        // ---- k = 1 .. G/2: rotate over partner lanes ----
        const int lane = threadIdx.x & 63;
        const int src = lane - gl + ((gl + G - 1) & (G - 1));
        // One rotation step of pass H (own particles H*PA .. H*PA+PA-1);
        // LAST is a compile-time flag: the final half step (k = G/2) visits
        // every pair from both sides, so each side only updates its own
        // particle and the lower half of the lanes tallies.
#define QMC_KSTEP(H, k, LAST)                                                 \
        {                                                                     \
            const bool count_pair = !(LAST) || gl < G / 2;                    \
            /* partner-major order: one partner table live at a time */       \
            _Pragma("unroll")                                                 \
            for (int b = 0; b < P; ++b) {                                     \
                PTab pb;                                                      \
                const int idx = (DUP == 2) ? b * 2 * G + gl + G - (k)         \
                                           : b * G + ((gl - (k)) & (G - 1));  \
                pb.s = lS[idx]; pb.c = lC[idx];                               \
                pb.su = lSU[idx]; pb.cu = lCU[idx];                           \
                const double pz = ZCLASS ? lZ[idx] : 0.0;                     \
                int pl = gl - (k); if (pl < 0) pl += G;                       \
                const bool pok = !PAD || (pl + G * b) < n;                    \
                _Pragma("unroll")                                             \
                for (int a = 0; a < PA; ++a) {                                \
                    constexpr int ao_base = (H) * PA;                         \
                    double q, Y; bool sh; unsigned long long shm;            \
                    if (FOURCASE)                                             \
                        pair_core4<ZCLASS>(pc, t[a], st4[FOURCASE ? a : 0],   \
                                           aks[a], akc[a], z[ao_base + a],    \
                                           pb, pz, q, Y, sh, shm);            \
                    else                                                      \
                        pair_core<ZCLASS>(pc, t[a], aks[a], akc[a],           \
                                          z[ao_base + a], pb, pz, q, Y, sh,   \
                                          shm);                               \
                    /* G = 64: the lower half of the lanes is bits 0..31 */   \
                    if (WAVE_COUNT)                                           \
                        ns_wave += __popcll((LAST) ? (shm & 0xffffffffull)    \
                                                   : shm);                    \
                    if (!PAD || (ok[ao_base + a] && pok)) {                   \
                        F[ao_base + a] += q;                                  \
                        if (!(LAST)) T[b] -= q;                               \
                        if (!(LAST) || count_pair) { QMC_TALLY(q, Y, sh); }   \
                        if (ITH) {                                            \
                            double kk = QMC_PAIR_KIN(q, sh);                  \
                            Kown[ao_base + a] += kk;                          \
                            if (!(LAST)) KT[b] += kk;                         \
                        }                                                     \
                    }                                                         \
                }                                                             \
                /* large P: keep the scheduler from interleaving the pairs   \
                   of different partners (it trades the occupancy away       \
                   for it: 282 registers instead of ~180 at P = 8) */        \
                if (P >= 4) __builtin_amdgcn_sched_barrier(0);                \
            }                                                                 \
            if (!(LAST)) {                                                    \
                _Pragma("unroll")                                             \
                for (int b = 0; b < P; ++b) {                                 \
                    T[b] = __shfl(T[b], src, 64);                             \
                    if (ITH) KT[b] = __shfl(KT[b], src, 64);                  \
                }                                                             \
            }                                                                 \
            /* P = 1: 16 factors between folds, each >= sin(pi rm / L) or   \
               cos(k2 rm - phi): no underflow for any admissible model */     \
            if (WF && (((k) & 15) == 0 || P > 1)) {                           \
                QMC_FOLD(prodS, expS);                                        \
                QMC_FOLD(prodL, expL);                                        \
            }                                                                 \
        }
#define QMC_PASS(H)                                                           \
        if ((H) < NPASS) {                                                    \
            if (NPASS > 1) {                                                  \
                _Pragma("unroll")                                             \
                for (int a = 0; a < PA; ++a) {                                \
                    QMC_LOAD_OWN(t[a], (H) * PA + a);                         \
                    aks[a] = m.a_long * t[a].s;                               \
                    akc[a] = m.a_long * t[a].c;                               \
                }                                                             \
            }                                                                 \
            /* (kept rolled: unrolled, the scheduler hoists the LDS reads of \
               every copy and the kernel loses half its occupancy: -8 %) */  \
            QMC_SECTION("rotation_loop_body");                                \
            for (int k = 1; k < G / 2; ++k)                                   \
                QMC_KSTEP(H, k, false)                                        \
            QMC_SECTION("rotation_last_step");                                \
            QMC_KSTEP(H, G / 2, true)                                         \
            /* deliver the travelling sums to their owners (lane gl ^ G/2    \
               holds them) and start the next pass from zero */              \
            _Pragma("unroll")                                                 \
            for (int b = 0; b < P; ++b) {                                     \
                F[b] += __shfl_xor(T[b], G / 2, 64);                          \
                T[b] = 0.0;                                                   \
                if (ITH) {                                                    \
                    Kown[b] += __shfl_xor(KT[b], G / 2, 64);                  \
                    KT[b] = 0.0;                                              \
                }                                                             \
            }                                                                 \
        }
        QMC_PASS(0)
        QMC_PASS(1)
#undef QMC_PASS
#undef QMC_KSTEP
    }

    // ---- local energy ----
    QMC_SECTION("energy+logwf");
    double e_lane = 0.0;
    if (ITH) {
#pragma unroll
        for (int a = 0; a < P; ++a) {
            double e = ok[a] ? (Kown[a] + kin1[a] - F[a] * F[a]) : 0.0;
            eith[a] = e;
            e_lane += e;
        }
    } else {
        // sum over unordered pairs of (k2^2 + q^2) [short] and
        // (b_long + q^2 / beta) [long], counted for both partners
        int nlong = npair - nshort;
        double pk = Qs + (Qall - Qs) * m.inv_beta;
        if (!WAVE_COUNT)
            pk += m.k2sq * (double)nshort + m.b_long * (double)nlong;
        e_lane = 2.0 * pk;
        e_lane += kin1_sum;
#pragma unroll
        for (int a = 0; a < P; ++a)
            if (ok[a]) e_lane -= F[a] * F[a];
    }
    E = group_sum<G>(e_lane);
    if (WAVE_COUNT && !ITH && !m.is_ideal) {
        int nl_wave = n * (n - 1) / 2 - ns_wave;
        E += 2.0 * (m.k2sq * (double)ns_wave + m.b_long * (double)nl_wave);
    }
    if (WF) {
        const double LN2 = 0.693147180559945309417;
        // prodL holds every pair's |Y|, prodS the short ones (cos > 0):
        // long product = prodL / prodS (mantissas; exponents kept apart)
        double lw = log_pos(prod1 * prodS) +
                    m.beta * log_pos(fabs(fast_div(prodL, prodS))) +
                    LN2 * ((double)(expS + exp1) +
                           m.beta * (double)(expL - expS)) -
                    xoff_sum;
        if (!WAVE_COUNT) lw += (double)nshort * m.log_am;
        logwf = group_sum<G>(lw);
        if (WAVE_COUNT) logwf += (double)ns_wave * m.log_am;
    }
#undef QMC_FOLD
#undef QMC_LOAD_OWN
#undef QMC_TALLY
#undef QMC_PAIR_KIN
}
