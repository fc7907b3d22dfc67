// qmc_sorted64.h -- the pair sum of the benchmark shape (one walker per
// wavefront, one particle per lane, N = 64, pairs classified from the sines)
// on lanes held in EXACTLY ascending position.
//
// Why a second path.  PMC + micro-benchmarks of round 3 (profiles/r03_*): on
// gfx950 the scalar pipe issues one instruction per ~4.2 cycles and SIMD, the
// same as the fp64 vector pipe, and the two overlap only as far as the eight
// wavefronts of a SIMD happen to want different pipes.  The rotation loop of
// eval_walker executes 26-28 scalar instructions and branches per step next to
// 14-23 vector ones -- exec-mask regions, validity tests of the "leading
// steps", fold checks, short-pair counts, loop control -- and the kernel sat at
// vector 94 % / scalar 53-68 % busy with every instruction removed from one
// pipe reappearing as queueing behind the other.  The validity tests exist
// because the lanes were only approximately ordered (one transposition pass
// every fourth step).  With the lanes exactly ascending, and the lower copy of
// the LDS tables holding the particles one period below (pair_core1), the
// separation D'_k = z_lane - z'_(lane-k) seen at rotation step k is >= 0 and
// grows with k for every lane.  Hence, with D'_(G/2) < L - rm checked once per
// walker (no pair is short through the periodic image on the far side):
//   * a pair is short iff D' < rm, always in the single (unwrapped, ordered)
//     case: no generic branch, no sign tests;
//   * the leading steps (every lane's partner short) need ONE compare per
//     trip of two steps -- the farther partner against z - rm decides for
//     both -- instead of three per step plus a scalar reduction;
//   * once no lane had a short pair in a step none will later: the trailing
//     steps run without compare, ballot, count and exec-masked region;
//   * |a_m| is folded into the own short-range table, so log|psi| needs no
//     count of short pairs.
// Three loops, two steps per trip: leading (all short), general (classified
// lane by lane), trailing (all long).  In the stationary ensemble of the
// benchmark box they run 15 / 6 / 9 of the 31 full steps.  The general loop
// requests the partner's tables a trip ahead; the leading and the trailing
// loops read them in the trip (round 4: requested ahead they cost register
// moves at every back edge, and the round trip is the other wavefronts' to
// hide -- the kernel is bound by vector issue).  A step is 3.5-12 vector
// instructions per lane.
// Walkers that fail the once-per-walker checks (practically never: it takes 32
// particles inside L/4) take eval_walker, which is exact for any order.
#pragma once

#include "qmc_device.h"

#ifndef QMC_SORTED64
#define QMC_SORTED64 1
#endif
#ifndef QMC_LDS_AHEAD
#define QMC_LDS_AHEAD 0
#endif
// The partner's share of a pair's drift: added into an LDS row at the partner's
// table index with ds_add_f64 (1), or carried in a travelling register that
// rotates one lane per step (0: a subtraction and two DPP moves per step).
// Measured twice now (round 2 on the general path, round 3 here): the atomics
// remove 3 vector instructions per step of the energy pass and are SLOWER --
// VMC 1.882 against 1.829 ms, DMC 0.526 against 0.515 (profiles/
// r03_ab_variants.txt): ds_add_f64 occupies the LDS pipe for 32 cycles.
#ifndef QMC_T_LDS
#define QMC_T_LDS 0
#endif
// Long-range pairs of the energy-only evaluation (the DMC step: no log|psi|)
// from ONE table per particle, t = cot(pi z / L):
//   a_long cot(pi D' / L) = a_long (t_i t_j + 1) / (t_j - t_i),
// numerator and denominator in one instruction each instead of two (mul + fma)
// from the sin / cos tables -- two fp64 operations fewer per long pair, and the
// round-4 micro-benchmark (tools/ubench5.hip) says it is fp64 operations, not
// issue slots, that the chip's power budget pays for.  The pair class comes
// from the positions (partner above z - rm: the test the leading steps use)
// instead of from the sine, so the partner's second table entry is its position
// -- the same two LDS reads per general step.  cot is periodic in L, so the copy
// "one period below" is the same number; at z -> 0 it diverges: sin is clamped
// from below (a particle closer than 1e-290 to the seam), which changes a long
// pair's quotient by 1e-280 of itself, and a pair of two such particles is
// never long.  Accuracy otherwise that of the two-table form (the quotient of
// two correctly rounded operations on correctly rounded cotangents).
// The log|psi| pass needs sin(pi D' / L) itself (the factor of the product),
// so that pass keeps the sin / cos tables.
#ifndef QMC_COT
#define QMC_COT 1
#endif
// The energy pass of the two-pass VMC step (REUSE: tables of this
// configuration are in LDS) uses it too: every lane replaces its entries of the
// sine row by the cotangent (one division per accepted move) and the loops take
// the partner's position from the position row.
#ifndef QMC_COT_REUSE
#define QMC_COT_REUSE 1
#endif
// The short-range pairs of the same passes, likewise from one number per
// particle: with A = k2 z_i - phi (own) and B = k2 z'_j (partner),
//   f2'/f2 = -k2 tan(A - B) = -k2 (tan A - tan B) / (1 + tan A tan B):
// the own table holds u = kappa tan A and w = tan A / kappa, the partner's row
// v = kappa tan B (kappa = -k2 / a_long, the unit of the quotients), and
//   X = u_i - v_j,  Y = 1 + w_i v_j
// are one fp64 instruction each instead of two, from ONE partner entry instead
// of two (the cosine row of the k2-table is not used either).  tan has poles
// inside the box; the identity does not care -- a huge tan A against an
// ordinary tan B gives -kappa / tan B with the relative error of the operands --
// and two huge ones cannot meet in a short pair: its angle A - B lies in
// (-phi, k2 rm - phi), inside (-pi/2, 0) and bounded away from 0 by the model
// (the trial function rises all the way to rm).  The denominators of the
// per-particle tangents are clamped away from zero (sign kept) so that no
// product overflows.  Accuracy: the reference's golden configurations through
// this form, tests/test_gpu_sorted_pins.py, 2e-11.
#ifndef QMC_TAN
#define QMC_TAN 1
#endif

// Trailing rotation steps without classification (one particle per lane).  In
// the stationary ensemble of the benchmark box 15 of the 31 steps are
// all-short leading steps, 6 have both classes in the wavefront and 9 are
// all-long; round 3 had measured no gain from a separate trailing loop -- on
// ensembles 320 steps after a random start, where 14 steps were mixed and 5
// all-long (profiles/r04_ab_variants.txt section 13).
// One logarithm per lane in the log|psi| pass (pairs of lanes share the two
// products; eval_sorted64).
#ifndef QMC_LW_PAIRS
#define QMC_LW_PAIRS 1
#endif
#ifndef QMC_S64_TRAIL
#define QMC_S64_TRAIL 1
#endif
// The log|psi| pass with the pair class from the positions (QMC_WF_ZCLASS): a
// long lane computes the sine and multiplies it into the product of the long
// factors, a short lane computes the short-range factor and multiplies it into
// the product of the short ones -- the same instructions as with the class
// from the sine (which every lane computed first), but each on the lanes that
// need it only: fewer fp64 operations on live lanes, one more LDS read per step.
#ifndef QMC_WF_ZCLASS
#define QMC_WF_ZCLASS 0
#endif
template <bool WF, bool EN, bool REUSE>
struct SortedCot {
    static constexpr bool ON = QMC_COT && EN && !WF &&
                               (QMC_COT_REUSE || !REUSE);
    static constexpr bool TAN = ON && QMC_TAN;
    // The DMC step (energy only, tables built here) then uses THREE rows --
    // cotangent, kappa tan(k2 z), position -- and keeps them together: 4.7 KB
    // per walker at N = 128 instead of 7.8 (the VMC step needs all five rows
    // for its log|psi| pass and keeps the five-row layout in both passes).
    static constexpr bool COMPACT = TAN && !REUSE;
    static constexpr int ROW_SU = COMPACT ? 1 : 2;
    static constexpr int ROW_Z = COMPACT ? 2 : 4;
    static constexpr int ROWS = COMPACT ? 3 : 5;
};

// x / y with |y| kept above `tiny` (sign of y kept; y = 0 counts as positive)
__device__ __forceinline__ double div_clamped(double x, double y, double tiny)
{
    const double ya = fmax(__builtin_fabs(y), tiny);
    return fast_div(x, __builtin_copysign(ya, y));
}

// lane i takes the value of lane i - 1, lane 0 takes 0.0 (bound_ctrl: no
// copy of the old value first -- one v_mov_b32_dpp per word instead of two
// instructions; positions are >= 0, so lane 0 never reads as "below")
__device__ __forceinline__ double wave_shr1_f64(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(
        __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true),
        __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true));
}

// lanes (of the first nl) whose particle lies BELOW the particle of the lane
// before them: bit i set = the pair (i - 1, i) is out of order; 0 = ascending
__device__ __forceinline__ unsigned long long lanes_inverted64(double z,
                                                               int nl = 64)
{
    const bool live = (int)(threadIdx.x & 63) < nl;
    return __builtin_amdgcn_ballot_w64(live & (z < wave_shr1_f64(z)));
}

// true iff the first nl lanes hold ascending positions (wave-uniform)
__device__ __forceinline__ bool lanes_ascending64(double z, int nl = 64)
{
    return lanes_inverted64(z, nl) == 0ull;
}

// One compare-exchange pass: every lane against the lane at byte address
// `addr` (its partner of the pass, or itself), the lower lane of a pair keeping
// the smaller position.  `flip` is the sign bit for the upper lanes: with
// d = z_partner - z, the lower lane takes the partner's particle if d < 0, the
// upper one if d > 0, i.e. if -d < 0.  (6 vector instructions and 3
// ds_bpermute; written with a select per lane it compiled to 20.)
__device__ __forceinline__ void cmpxchg_pass64(double &z, int &lab, int addr,
                                               int flip)
{
    const int plo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(z));
    const int phi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(z));
    const int pl = __builtin_amdgcn_ds_bpermute(addr, lab);
    const double zp = __hiloint2double(phi, plo);
    const double d = zp - z;
    const bool take = __hiloint2double(__double2hiint(d) ^ flip,
                                       __double2loint(d)) < 0.0;
    z = take ? zp : z;
    lab = take ? pl : lab;
}

// Odd-even transposition passes until the lanes are ascending (a VMC / DMC move
// displaces a particle by a few per cent of the spacing: most steps need none
// or one double pass).  `anchor_seam` first: a particle that crossed the box
// boundary sits at the wrong end of the row and is rotated into place instead
// of being bubbled through 63 lanes.  Bounded: odd-even transposition sorts n
// items in n passes.
// (nl: lanes holding particles; the lanes above them never take part)
// The mask of the inverted pairs steers the work: the even pass exchanges the
// pairs (0,1)(2,3).. -- inversion bits at odd lanes -- the odd pass the pairs
// (1,2)(3,4).. -- bits at even lanes; a pass runs only if one of its pairs is
// inverted, and the seam test (`anchor_seam`: four readlanes) only if the
// inversion sits at an end of the row, where a particle that crossed the box
// boundary shows up.  (Both passes, the seam test and a re-check every trip
// cost 52 vector instructions per VMC step at 1.3 trips on average.)
__device__ __forceinline__ bool sort_lanes64(double &z, int &lab, int gl,
                                             int nl = 64)
{
    unsigned long long inv = lanes_inverted64(z, nl);
    if (inv == 0ull) return true;
    const int odd = gl & 1;
    const int pe = gl ^ 1;
    const int addr_even = ((gl >= nl || pe >= nl) ? gl : pe) << 2;
    const int po = odd ? (gl + 1) : (gl - 1);
    const int addr_odd = ((gl >= nl || po < 0 || po >= nl) ? gl : po) << 2;
    const int flip_even = odd << 31;            // the odd lane is the upper one
    const int flip_odd = (odd ^ 1) << 31;       // the even lane is
    const unsigned long long ends = 2ull | (1ull << (nl - 1));
    for (int it = 0; it < 140; ++it) {
        if (inv & ends) anchor_seam(z, lab, nl);
        if (inv & 0xAAAAAAAAAAAAAAAAull)
            cmpxchg_pass64(z, lab, addr_even, flip_even);
        if (inv & 0x5555555555555555ull)
            cmpxchg_pass64(z, lab, addr_odd, flip_odd);
        inv = lanes_inverted64(z, nl);
        if (inv == 0ull) return true;
    }
    return false;
}

// The once-per-walker condition beside the order: the farthest partner of the
// rotation (step G/2, lane ^ 32) is closer than L - rm, i.e. no pair of the 32
// steps is short-range through the image on the far side.
__device__ __forceinline__ bool far_partner_ok64(const DevModel &m, double z,
                                                 int gl)
{
    const double zp = __shfl_xor(z, 32, 64);
    // D' = z - zp for the upper half of the lanes, z - (zp - L) for the lower
    const double d = (gl < 32) ? (z - zp) + m.L : z - zp;
    return __builtin_amdgcn_ballot_w64(d >= m.L_minus_rm) == 0ull;
}

// The same for a row of nl < 64 lanes: the partner of the last step, nl / 2
// lanes down the ring.
__device__ __forceinline__ bool far_partner_ok_ring(const DevModel &m, double z,
                                                    int gl, int nl)
{
    const int K = nl / 2;
    const bool live = gl < nl;
    int src = gl - K;
    const bool wrapped = src < 0;
    if (wrapped) src += nl;
    const double zp = __shfl(z, live ? src : gl, 64);
    const double d = wrapped ? (z - zp) + m.L : z - zp;
    return __builtin_amdgcn_ballot_w64(live & (d >= m.L_minus_rm)) == 0ull;
}

// A table read that stays where it is written: the loops below request the
// partner's entries ahead of their use, and the optimizer otherwise sinks a
// plain load back to its first use (the round trip then lies in the step's
// critical path again).
template <typename R>
__device__ __forceinline__ R lds_ahead(const R *p)
{
#if QMC_LDS_AHEAD
    // (an explicit LDS pointer: a volatile access through a generic one is
    // issued as a flat load)
    typedef const volatile __attribute__((address_space(3))) R *lds_ptr;
    return *(lds_ptr)p;
#else
    return *p;
#endif
}

template <typename R>
struct Own64 {
    R s, c;          // sin, cos(pi z / L)
    R s0, c0;        // |a_m| sin, cos(k2 z - phi): Y = f2 itself for a short pair
    // (-k2 / a_long) times them: the short-range numerator.  The pair quotients
    // are summed in units of a_long = (pi / L) beta -- a long pair's is plain
    // cot(pi D' / L), numerator c_i c_j + s_i s_j from the tables as they are --
    // and the sums are scaled once at the end (F = f1'/f1 + a_long sum q,
    // sum q^2 by a_long^2): two table registers fewer per particle than with
    // a_long folded into a second copy of (s, c)
    R ks0, kc0;
    R zt;            // z - rm: a partner above it is closer than rm
};

// The one-body part of a particle (as in eval_walker).
struct SortedOneBody {
    double ldz = 0.0;     // f1'/f1
    double kin1 = 0.0;    // -f1''/f1 + ldz^2 + V; ldz^2 alone when every barrier
                          // is alike (the region constants are then counted
                          // per wavefront from `barrier`)
    double xoff = 0.0;    // -log f1 (WF)
    double prod1 = 1.0;   // f1 itself on the direct path (WF)
    bool barrier = false;
};

// Everything one particle (slot `slot` of NS, position z) contributes before
// the pair loop: its one-body factor, its own pair tables `o`, and -- unless
// the tables of this configuration are already there (REUSE) -- its entries of
// the LDS tables: 5 rows (sin, cos(pi z / L), sin, cos(k2 z), z) of NS + NS/2
// (+ 1) entries: [H + slot] = the particle, [H + slot - n] = the particle one
// period below, kept for the upper slots only -- the rotation reaches NS/2
// slots down and no further (7.5 KB per walker at N = 128: five wavefronts per
// SIMD; the full doubled tables allowed four).
// `n`: slots in use (n = NS for the exact shapes; a slot >= n writes nothing).
// The rows carry one unused entry in front: the loops request tables up to one
// step past their last one.
// Two slots per lane (NS = 128): TWO unused entries in front, which keeps the
// pair (slot 2 l, slot 2 l + 1) of a lane on a 16-byte boundary in every row --
// the loops of qmc_sorted128.h read a partner lane's two entries with ONE
// ds_read_b128 (two ds_read_b64 at a lane stride of 16 bytes are 2-way bank
// conflicts: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE was 44 % in round 3) --
// and keeps the look-ahead of the last trip (k + 3 = N / 4 + 2 lanes down)
// inside the row.
template <int NS>
struct SortedRows {
    static constexpr int H = NS / 2 + (NS > 64 ? 2 : 1);   // offset of slot 0
    static constexpr int ROW = NS + H;
    static_assert(NS <= 64 || (H % 2 == 0 && ROW % 2 == 0),
                  "two slots per lane: 16-byte aligned pairs");
};

// What a particle publishes in the LDS rows: its entries of the first / second
// row and of the k2 rows, and the k2 entries of its image one period below.
struct SortedPub {
    double s, c, su, cu, su_lo, cu_lo;
};

// WRITE = false: the caller publishes the entries itself from `pub` (two slots
// per lane: both particles of a lane with one 16-byte store per row).
template <typename R, bool WF, bool EN, bool REUSE, int NS, bool WRITE = true>
__device__ __forceinline__ void sorted_particle_setup(const DevModel &m, double z,
                                                      int slot, R *tab,
                                                      Own64<R> &o,
                                                      SortedOneBody &ob,
                                                      int n, PTab &ta,
                                                      SortedPub &pub)
{
    constexpr int H = SortedRows<NS>::H, ROW = SortedRows<NS>::ROW;
    typedef SortedCot<WF, EN, REUSE> RowsOf;
    R *lS = tab, *lC = lS + ROW, *lSU = lS + RowsOf::ROW_SU * ROW,
      *lCU = lS + 3 * ROW, *lZ = lS + RowsOf::ROW_Z * ROW;
    TrigRow trow;
    const bool trig_ok = !REUSE && m.trig_table && trig_tab_load(m, z, trow);
    if (!m.is_free && m.ob_table) {
        double lf = 0.0;
        one_body_tab<WF, EN>(m, z, ob.ldz, lf, ob.barrier);
        if (WF) ob.xoff = -lf;
        if (EN) {
            if (m.uniform_barrier)
                ob.kin1 = ob.ldz * ob.ldz;
            else
                ob.kin1 = fma(ob.ldz, ob.ldz,
                              one_body_kin_const(m, z, ob.barrier));
        }
    } else if (!m.is_free) {
        double kp, f1, xo;
        one_body(m, z, ob.ldz, kp, f1, xo);
        if (EN) ob.kin1 = kp;
        if (WF) { ob.prod1 = f1; ob.xoff = xo; }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (REUSE) {
        ta.s = (double)lS[H + slot]; ta.c = (double)lC[H + slot];
        ta.su = (double)lSU[H + slot]; ta.cu = (double)lCU[H + slot];
    } else if (trig_ok) {
        trig_tab_finish(m, trow, ta);
    } else {
        sincos_halfpi(z * m.two_over_L, ta.s, ta.c);
        sincos_halfpi(z * m.k2_2pi, ta.su, ta.cu);
    }
    constexpr bool COT = SortedCot<WF, EN, REUSE>::ON;
    constexpr bool TAN = SortedCot<WF, EN, REUSE>::TAN;
    // what the partners read of this particle (first row, second row, k2 rows)
    pub.s = ta.s; pub.c = ta.c; pub.su = ta.su; pub.cu = ta.cu;
    const double s0 = fma(ta.su, m.am_cphi, -(ta.cu * m.am_sphi));
    const double c0 = fma(ta.cu, m.am_cphi, ta.su * m.am_sphi);
    // the k2 entry one period below: rotation by k2 L
    pub.su_lo = fma(ta.su, m.cth, -(ta.cu * m.sth_signed));
    pub.cu_lo = fma(ta.cu, m.cth, ta.su * m.sth_signed);
    if (COT && TAN) {
        // cot(pi z / L) = c / s, tan(k2 z - phi) = s0 / c0, tan(k2 z) = su / cu
        // and tan(k2 z - k2 L) = su_lo / cu_lo: FOUR quotients from ONE
        // reciprocal (of the product of the denominators; 15 fp64 operations
        // and one v_rcp_f64 instead of 20 and four).  Denominators are kept
        // away from zero (sign kept; 1e-70: the product of four of them and of
        // two of the quotients stays finite); the float pair loop clamps the
        // quotients into float's range as well.
        const double tiny = 1e-70;
        const double d0 = fmax(ta.s, tiny);            // (s >= 0 inside the box)
        const double d1 = __builtin_copysign(fmax(__builtin_fabs(c0), tiny), c0);
        const double d2 = __builtin_copysign(fmax(__builtin_fabs(ta.cu), tiny),
                                             ta.cu);
        const double d3 = __builtin_copysign(
            fmax(__builtin_fabs(pub.cu_lo), tiny), pub.cu_lo);
        const double p01 = d0 * d1, p23 = d2 * d3, pall = p01 * p23;
        double r = __builtin_amdgcn_rcp(pall);
        r = fma(fma(-pall, r, 1.0), r, r);
        r = fma(fma(-pall, r, 1.0), r, r);
        const double i01 = p23 * r, i23 = p01 * r;      // 1 / p01, 1 / p23
        double cotz = ta.c * (i01 * d1);
        double tA = s0 * (i01 * d0);
        double tB = ta.su * (i23 * d3);
        double tBl = pub.su_lo * (i23 * d2);
        if (sizeof(R) == 4) {
            const double big = 1e18;
            cotz = fmin(fmax(cotz, -big), big); tA = fmin(fmax(tA, -big), big);
            tB = fmin(fmax(tB, -big), big); tBl = fmin(fmax(tBl, -big), big);
        }
        pub.s = cotz;
        pub.c = z;      // (o.c: the position -- the partner's comes from lZ)
        o.s0 = (R)(m.m_k2_over_a * tA);                  // u = kappa tan A
        o.c0 = (R)(tA * m.m_a_over_k2);                  // w = tan A / kappa
        pub.su = m.m_k2_over_a * tB;                     // kappa tan(k2 z)
        pub.su_lo = m.m_k2_over_a * tBl;
    } else {
        // (clamp: the product of two clamped quantities stays finite)
        const double tiny = sizeof(R) == 4 ? 1e-18 : 1e-140;
        if (COT) {
            // (s >= 0 inside the box)
            pub.s = fast_div(ta.c, fmax(ta.s, tiny));    // cot(pi z / L)
            pub.c = z;
        }
        o.s0 = (R)s0; o.c0 = (R)c0;
        if (EN) {
            o.ks0 = (R)(m.m_k2_over_a * s0);
            o.kc0 = (R)(m.m_k2_over_a * c0);
        }
    }
    o.s = (R)pub.s; o.c = (R)pub.c;
    o.zt = (R)(z - m.rm);
    if (WRITE && !REUSE && slot < n) {
        // (COT: the first row holds the cotangent, the cosine row is not used;
        // TAN: the k2 sine row holds kappa tan(k2 z), its cosine row is not used)
        lS[H + slot] = (R)pub.s;
        if (!COT) lC[H + slot] = (R)pub.c;
        lSU[H + slot] = (R)pub.su;
        if (!TAN) lCU[H + slot] = (R)pub.cu;
        lZ[H + slot] = (R)z;
        // one period below: the entry the slots up to NS / 2 above the start
        // of the row find when they look past slot 0
        const int lo = H + slot - n;
        if (lo >= 1) {
            // (cot: the same number one period below)
            lS[lo] = COT ? (R)pub.s : (R)-pub.s;
            if (!COT) lC[lo] = (R)-pub.c;
            lSU[lo] = (R)pub.su_lo;
            if (!TAN) lCU[lo] = (R)pub.cu_lo;
            lZ[lo] = (R)(z - m.L);
        }
    } else if (WRITE && REUSE && COT && slot < n) {
        // the energy pass after an accepted VMC move: sine row -> cotangent
        // row, k2 sine row -> tangent row (every lane rewrites its own entries
        // only, and has read them above)
        lS[H + slot] = (R)pub.s;
        if (TAN) lSU[H + slot] = (R)pub.su;
        const int lo = H + slot - n;
        if (lo >= 1) {
            lS[lo] = (R)pub.s;
            if (TAN) lSU[lo] = (R)pub.su_lo;
        }
    }
}

// One walker on ascending lanes.  z: the lane's particle; lds: the 5 rows of
// sorted_particle_setup (97 entries each).
//   WF    : logwf out;   EN: E and F (drift of the lane's particle) out
//   REUSE : the tables of this configuration are already in LDS
//   PAD   : only the first nl = N < 64 lanes hold particles.  The ring then has
//           nl members: nl / 2 rotation steps, the last of them a half step
//           only if nl is even; the travelling sum moves through ds_bpermute
//           (no DPP rotation over a partial wavefront), and every wave-wide
//           test and count is restricted to the lanes in use.
template <typename R, bool WF, bool EN, bool REUSE, bool PAD = false>
__device__ __forceinline__ void eval_sorted64(const DevModel &m, double z, int gl,
                                              double *lds, double &F, double &E,
                                              double &logwf)
{
    constexpr int G = 64, H = SortedRows<G>::H, ROW = SortedRows<G>::ROW;
    constexpr bool COT = SortedCot<WF, EN, REUSE>::ON;
    constexpr bool TAN = SortedCot<WF, EN, REUSE>::TAN;
    constexpr bool ZW = QMC_WF_ZCLASS && WF && !EN;   // (PL = long factors only)
    // (1: in every pass; 2: in the passes that compute the energy)
    constexpr bool TRAIL = QMC_S64_TRAIL == 1 || (QMC_S64_TRAIL == 2 && EN);
    typedef SortedCot<WF, EN, REUSE> RowsOf;
    R *lS = (R *)lds, *lC = lS + ROW, *lSU = lS + RowsOf::ROW_SU * ROW,
      *lCU = lS + 3 * ROW, *lZ = lS + RowsOf::ROW_Z * ROW;
    // sixth row (double whatever R is): the sums the partners collect
    constexpr bool T_LDS = QMC_T_LDS && EN;
    double *lA = lds + RowsOf::ROWS * ROW;
    const int nl = PAD ? m.n : G;            // lanes in use = particles
    const int K = nl / 2;                    // rotation steps
    const bool half_last = !PAD || (nl & 1) == 0;   // step K is a half step
    const int kfull = half_last ? K - 1 : K; // steps that visit a pair once
    const bool live = !PAD || gl < nl;
    const unsigned long long live_mask =
        PAD ? __builtin_amdgcn_ballot_w64(live) : ~0ull;
    [[maybe_unused]] constexpr int QMC_SEC_OFF = REUSE ? QMC_NSEC / 2 : 0;
    QMC_SECTION("tables+onebody");
    Own64<R> o;
    SortedOneBody ob;
    PTab ta_own;
    SortedPub pub_own;
    sorted_particle_setup<R, WF, EN, REUSE, G>(m, z, gl, (R *)lds, o, ob, nl,
                                               ta_own, pub_own);
    // (where the shares of this lane's particle arrive: its own index from the
    // lanes above it, the index one period below from the lanes that reach it
    // around the end of the row)
    const int a_lo = H + gl - nl;
    if (T_LDS && live) {
        lA[H + gl] = 0.0;
        if (a_lo >= 1) lA[a_lo] = 0.0;
    }
    if (!REUSE || T_LDS || COT) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const double ldz = ob.ldz, kin1 = ob.kin1, xoff = ob.xoff,
                 prod1 = ob.prod1;
    int nb_wave = 0;
    const bool nb_counted = EN && !m.is_free && m.ob_table && m.uniform_barrier;
    if (nb_counted) nb_wave = __popcll(__ballot(ob.barrier & live));
    const R sin_rm = (R)m.sin_rm;
    // partner of rotation step k: entry (H + gl) - k
    // (COT: the second entry of a general step is the partner's position)
    const R *pS = lS + H + gl, *pC = (COT ? lZ : lC) + H + gl,
            *pSU = lSU + H + gl, *pCU = lCU + H + gl, *pZ = lZ + H + gl;
    typedef __attribute__((address_space(3))) double *lds_dptr;
    const lds_dptr pA = (lds_dptr)(lA + H + gl);
    // the lane below in the ring of the lanes in use (travelling sums)
    int ring_src = 0;
    if (PAD) ring_src = (live ? (gl == 0 ? nl - 1 : gl - 1) : gl) << 2;
#define QMC_S64_ROR(T)                                                        \
    (PAD ? __hiloint2double(                                                  \
               __builtin_amdgcn_ds_bpermute(ring_src, __double2hiint((double)(T))), \
               __builtin_amdgcn_ds_bpermute(ring_src, __double2loint((double)(T)))) \
         : (double)group_ror1<G>(T))

    R Fr = 0;                // drift: pair quotients, in units of a_long
    R T = 0;                 // travelling sum for the partner lane
    R Qall = 0, Qs = 0;      // sum of q^2 over all / short pairs
    R PS = 1, PL = 1;        // products: short factors f2, |Y| of all pairs
    int eS = 0, eL = 0;      // their binary exponents (float pair loop)
    int ns = 0;              // short pairs (EN: region constants of the energy)
    int k = 1;

    // ---- leading steps: every lane's partner is closer than rm ----
    // Two steps per trip with two sets of registers for the requested tables:
    // each set is refilled right after its step, two steps ahead of its next
    // use, with no copies and one address update per trip.  (Left to the
    // compiler: rolled, the loop carries two 64-bit moves and an address add per
    // step; unrolled by its constant trip count the kernel is 25 KB.)
    QMC_SECTION("leading_short_steps");
    // numerator / denominator of a leading (all-short) step
    // (TAN: bsu = the partner's kappa tan(k2 z'), bcu is not read)
#define QMC_S64_LEAD_XY(bsu, bcu, X, Y)                                       \
    const R Y = TAN ? q_fma(o.c0, (bsu), (R)1)      /* 1 + w_i v_j */          \
                    : o.c0 * (bcu) + o.s0 * (bsu);  /* f2 = |a_m| cos(k2 D' - phi) */ \
    R X = 0;                                                                  \
    if (EN) X = TAN ? o.s0 - (bsu)                  /* u_i - v_j */            \
                    : o.ks0 * (bcu) - o.kc0 * (bsu);                          \
    if (WF) PS *= Y;
    // the quotient's way into the sums of step kk (not the last step)
#define QMC_S64_ADD_Q(q, kk)                                                  \
    {                                                                         \
        Fr += (q);                                                            \
        if (T_LDS) {                                                          \
            if (live)                                                         \
                (void)__builtin_amdgcn_ds_atomic_fadd_f64(pA - (kk),          \
                                                          -(double)(q));      \
        } else {                                                              \
            T -= (q);                                                         \
            T = (R)QMC_S64_ROR(T);                                            \
        }                                                                     \
    }
    // every lane in use says yes
#define QMC_S64_ALL(cond)                                                     \
    ((__builtin_amdgcn_ballot_w64(cond) | ~live_mask) == ~0ull)
    {
        // (TAN: the cosine row of the k2-table is not read -- nor written)
        R asu = lds_ahead(pSU - 1), acu = TAN ? (R)0 : lds_ahead(pCU - 1),
          az = lds_ahead(pZ - 1);
        R bsu = lds_ahead(pSU - 2), bcu = TAN ? (R)0 : lds_ahead(pCU - 2),
          bz = lds_ahead(pZ - 2);
        // (k odd at the top; both steps of a trip are full steps.  The row
        // ascends: when the FARTHER partner of a trip is short for every lane,
        // so is the nearer one -- one wave-wide test per trip, on az no more)
#pragma clang loop unroll(disable)
        while (k < kfull) {
            // (a last all-short step whose successor is not is left to the
            // general steps: nothing of this trip is needed after the exit, and
            // the loop carries neither the old product nor the old address)
            if (!QMC_S64_ALL(bz > o.zt)) break;
            QMC_S64_LEAD_XY(asu, acu, Xa, Ya)
            asu = lds_ahead(pSU - (k + 2));
            if (!TAN) acu = lds_ahead(pCU - (k + 2));
            az = lds_ahead(pZ - (k + 2));
            QMC_S64_LEAD_XY(bsu, bcu, Xb, Yb)
            bsu = lds_ahead(pSU - (k + 3));
            if (!TAN) bcu = lds_ahead(pCU - (k + 3));
            bz = lds_ahead(pZ - (k + 3));
            k += 2;
            if (EN) {
                // (one reciprocal for both quotients -- r = 1 / (Ya Yb), qa = Xa r
                // Yb -- measured no faster, 1 % slower in the VMC step: the
                // reciprocal does not hold up the multiply-add pipe)
                const R qa = pair_div(Xa, Ya), qb = pair_div(Xb, Yb);
                QMC_S64_ADD_Q(qa, k - 2)
                QMC_S64_ADD_Q(qb, k - 1)
                Qs = q_fma(qa, qa, Qs);
                Qs = q_fma(qb, qb, Qs);
            }
            if (WF && sizeof(R) == 4 && (k & 7) == 1) q_fold(PS, eS);
        }
    }
#undef QMC_S64_LEAD_XY
    // (these pairs belong to both products and both sums)
    if (WF && !ZW) { PL = PS; eL = eS; }
    if (EN) { Qall = Qs; ns = (k - 1) * nl; }

    // ---- general steps: classified pair by pair ----
    QMC_SECTION("rotation_loop_body");
    // numerator / denominator / class of a general step
    // (COT: cs = the partner's cotangent, cc = its position)
    // (ZW: the log|psi| pass with the class from the partner's position zj)
#define QMC_S64_ZW(cs, cc, zj, kk, LAST)                                      \
    {                                                                         \
        const bool mine = live & (!(LAST) || gl < K);                         \
        const bool sh = (zj) > o.zt;                                          \
        if (!sh) {                                                            \
            asm volatile("");                                                 \
            const R Y = o.s * (cc) - o.c * (cs);                              \
            if (mine) PL *= Y;                                                \
        }                                                                     \
        if (sh) {                                                             \
            asm volatile("");                                                 \
            const R bsu_ = pSU[-(kk)], bcu_ = pCU[-(kk)];                     \
            const R Y = o.c0 * bcu_ + o.s0 * bsu_;                            \
            if (mine) PS *= Y;                                                \
        }                                                                     \
    }
#define QMC_S64_XY(cs, cc, kk, LAST, X, Y, sh, mine)                          \
    const R Y##_s = COT ? (cs) - o.s           /* t_j - t_i */                 \
                        : o.s * (cc) - o.c * (cs);  /* sin(pi D' / L) >= 0 */  \
    R X = 0;                                                                  \
    if (EN) X = COT ? q_fma(o.s, (cs), (R)1)   /* t_i t_j + 1 */               \
                    : o.c * (cc) + o.s * (cs); /* cos(pi D' / L) */            \
    const bool mine = live & (!(LAST) || gl < K);                             \
    const bool sh = COT ? (cc) > o.zt          /* D' < rm */                   \
                        : q_abs(Y##_s) < sin_rm;                              \
    /* (the lanes with a short pair, taken where the compare is: asked for    \
       after the exec-masked region the mask is rebuilt from a select) */     \
    const unsigned long long sh##_m =                                         \
        __builtin_amdgcn_ballot_w64(sh & mine);                               \
    if (EN) ns += __popcll(sh##_m);                                           \
    R Y = Y##_s;                                                              \
    if (sh) {                                                                 \
        asm volatile("");                      /* exec-masked, not selects */  \
        const R bsu_ = pSU[-(kk)];                                            \
        if (TAN) {                                                            \
            Y = q_fma(o.c0, bsu_, (R)1);                                      \
            X = o.s0 - bsu_;                                                  \
        } else {                                                              \
            const R bcu_ = pCU[-(kk)];                                        \
            Y = o.c0 * bcu_ + o.s0 * bsu_;                                    \
            if (EN) X = o.ks0 * bcu_ - o.kc0 * bsu_;                          \
        }                                                                     \
        if (WF && mine) PS *= Y;                                              \
    }                                                                         \
    if (WF && mine) PL *= Y;
    // q^2 into the sums over all / short pairs
#define QMC_S64_TALLY(q, sh, mine)                                            \
    if (mine) {                                                               \
        Qall = q_fma(q, q, Qall);                                             \
        if (sh) {                                                             \
            asm volatile("");                                                 \
            Qs = q_fma(q, q, Qs);                                             \
        }                                                                     \
    }
    if constexpr (ZW) {
        R as_ = lds_ahead(pS - k), ac_ = lds_ahead(pC - k), az_ = lds_ahead(pZ - k);
        R bs_ = lds_ahead(pS - (k + 1)), bc_ = lds_ahead(pC - (k + 1)),
          bz_ = lds_ahead(pZ - (k + 1));
#pragma clang loop unroll(disable)
        while (k < kfull) {
            QMC_S64_ZW(as_, ac_, az_, k, false)
            as_ = lds_ahead(pS - (k + 2)); ac_ = lds_ahead(pC - (k + 2));
            az_ = lds_ahead(pZ - (k + 2));
            QMC_S64_ZW(bs_, bc_, bz_, k + 1, false)
            bs_ = lds_ahead(pS - (k + 3)); bc_ = lds_ahead(pC - (k + 3));
            bz_ = lds_ahead(pZ - (k + 3));
            k += 2;
            if (sizeof(R) == 4) {
                q_fold(PS, eS);
                q_fold(PL, eL);
            }
        }
        if (k <= kfull) {
            QMC_S64_ZW(as_, ac_, az_, k, false)
            ++k;
            as_ = bs_; ac_ = bc_; az_ = bz_;
        }
        QMC_SECTION("rotation_last_step");
        if (half_last) QMC_S64_ZW(as_, ac_, az_, k, true)
    } else {
        R as_ = lds_ahead(pS - k), ac_ = lds_ahead(pC - k);   // step k
        // step k + 1 (<= K + 1: inside the rows)
        R bs_ = lds_ahead(pS - (k + 1)), bc_ = lds_ahead(pC - (k + 1));
#pragma clang loop unroll(disable)
        while (k < kfull) {
            QMC_S64_XY(as_, ac_, k, false, Xa, Ya, sha, minea)
            as_ = lds_ahead(pS - (k + 2)); ac_ = lds_ahead(pC - (k + 2));
            QMC_S64_XY(bs_, bc_, k + 1, false, Xb, Yb, shb, mineb)
            // (k + 3 <= K + 2: the unused entry in front of the rows)
            bs_ = lds_ahead(pS - (k + 3)); bc_ = lds_ahead(pC - (k + 3));
            k += 2;
            if (EN) {
                const R qa = pair_div(Xa, Ya), qb = pair_div(Xb, Yb);
                QMC_S64_ADD_Q(qa, k - 2)
                QMC_S64_ADD_Q(qb, k - 1)
                QMC_S64_TALLY(qa, sha, minea)
                QMC_S64_TALLY(qb, shb, mineb)
            }
            if (WF && sizeof(R) == 4) {
                q_fold(PS, eS);
                q_fold(PL, eL);
            }
            // the row ascends: a lane's partners only get farther, so once
            // no lane had a short pair in a step none will in a later one
            if (TRAIL && shb_m == 0ull) break;
        }
        // ---- trailing steps: every pair is long-range, no classification ----
        if (TRAIL) {
            QMC_SECTION("rotation");
#define QMC_S64_LONG(cs, cc)                                                  \
            {                                                                 \
                const R Y = COT ? (cs) - o.s : o.s * (cc) - o.c * (cs);       \
                if (WF && live) PL *= Y;                                      \
                if (EN) {                                                     \
                    const R X = COT ? q_fma(o.s, (cs), (R)1)                  \
                                    : o.c * (cc) + o.s * (cs);                \
                    const R q = pair_div(X, Y);                               \
                    QMC_S64_ADD_Q(q, k)                                       \
                    if (live) Qall = q_fma(q, q, Qall);                       \
                }                                                             \
                ++k;                                                          \
            }
#pragma clang loop unroll(disable)
            while (k < kfull) {
                // (the entries of a trip are read IN the trip: requested a
                // trip ahead they arrive in other registers than the loop
                // carries and cost two 64-bit moves per trip -- the latency is
                // the other wavefronts' to hide, the moves are nobody's)
                QMC_S64_LONG(pS[-k], pC[-k])
                QMC_S64_LONG(pS[-k], pC[-k])
                if (WF && sizeof(R) == 4) q_fold(PL, eL);
            }
            as_ = pS[-k]; ac_ = pC[-k];
            bs_ = pS[-(k + 1)]; bc_ = pC[-(k + 1)];
#undef QMC_S64_LONG
        }
        if (k <= kfull) {
            // an odd number of full steps was left: the last one is in the
            // first set
            QMC_S64_XY(as_, ac_, k, false, Xa, Ya, sha, minea)
            if (EN) {
                const R q = pair_div(Xa, Ya);
                QMC_S64_ADD_Q(q, k)
                QMC_S64_TALLY(q, sha, minea)
            }
            ++k;
            as_ = bs_; ac_ = bc_;
        }
        // the final half step (an even number of lanes in use) visits every
        // pair from both sides: each side updates its own particle, the lower
        // half of the lanes tallies
        QMC_SECTION("rotation_last_step");
        if (half_last) {
            QMC_S64_XY(as_, ac_, k, true, Xl, Yl, shl, minel)
            if (EN) {
                const R q = pair_div(Xl, Yl);
                Fr += q;
                QMC_S64_TALLY(q, shl, minel)
            }
        }
    }
#undef QMC_S64_XY
#undef QMC_S64_ZW
#undef QMC_S64_TALLY
#undef QMC_S64_ADD_Q
#undef QMC_S64_ALL
    if (T_LDS) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double t = lA[H + gl];
        if (a_lo >= 1) t += lA[a_lo];
        Fr += (R)t;
    } else if (EN) {
        // after kfull rotations lane l holds the sum of particle l - kfull - 1
        if (PAD) {
            int src = gl + kfull + 1;
            if (src >= nl) src -= nl;
            Fr += __shfl(T, live ? src : gl, 64);
        } else {
            Fr += __shfl_xor(T, G / 2, 64);
        }
    }
#undef QMC_S64_ROR

    QMC_SECTION("energy+logwf");
    double e_lane = 0.0, e_consts = 0.0;
    if (EN) {
        F = fma(m.a_long, (double)Fr, ldz);
        const double Qall_d = (double)Qall, Qs_d = (double)Qs;
        const double pk = (Qs_d + (Qall_d - Qs_d) * m.inv_beta) * m.a_long_sq;
        e_lane = fma(2.0, pk, kin1) - F * F;
        if (PAD && !live) e_lane = 0.0;
        if (nb_counted)
            e_consts += (double)(nl - nb_wave) * m.e0 +
                        (double)nb_wave * (m.v_barrier - m.v0_minus_e0);
        const int nlong = nl * (nl - 1) / 2 - ns;
        e_consts += 2.0 * (m.k2sq * (double)ns + m.b_long * (double)nlong);
    }
    double lw = 0.0;
    if (WF) {
        const double LN2 = 0.693147180559945309417;
        const double PS_d = (double)PS, PL_d = (double)PL;
        if (ZW) {
            // (PL: the long factors alone)
            lw = fma(m.beta, log_pos(PL_d),
                     log_pos((!m.is_free && m.ob_table) ? PS_d : prod1 * PS_d));
        } else if (QMC_LW_PAIRS && sizeof(R) == 8 && !m.is_free &&
                   m.ob_table) {
            // ONE logarithm per lane instead of two.  What the sum over the
            // lanes needs is beta sum log PL + (1 - beta) sum log PS, and a
            // wavefront pays for a logarithm per instruction sequence, not per
            // lane: the lanes of a pair (2i, 2i + 1) exchange one product each,
            // the even lane takes beta log(PL PL'), the odd one (1 - beta)
            // log(PS PS').  (Products of 2 x 63 factors of the order of one:
            // far inside the range of a double.)
            const bool odd = gl & 1;
            double own = odd ? PS_d : PL_d, give = odd ? PL_d : PS_d;
            if (PAD && !live) { own = 1.0; give = 1.0; }
            // quad_perm [1, 0, 3, 2]: the other lane of the pair
            const double got = __hiloint2double(
                __builtin_amdgcn_update_dpp(0, __double2hiint(give), 0xB1, 0xf,
                                            0xf, false),
                __builtin_amdgcn_update_dpp(0, __double2loint(give), 0xB1, 0xf,
                                            0xf, false));
            // (the two coefficients as scalars: left alone, the compiler selects
            // an ADDRESS per lane and loads the coefficient from the model in
            // global memory)
            int bh = __double2hiint(m.beta), bl = __double2loint(m.beta);
            int oh = __double2hiint(m.one_minus_beta),
                ol = __double2loint(m.one_minus_beta);
            asm volatile("" : "+s"(bh), "+s"(bl), "+s"(oh), "+s"(ol));
            lw = __hiloint2double(odd ? oh : bh, odd ? ol : bl) *
                 log_pos(own * got);
            // (an idle lane next to the last lane in use carries that lane's
            // short-range part; idle pairs carry nothing)
            if (PAD && gl >= ((nl + 1) & ~1)) lw = 0.0;
            if (!PAD || live) lw -= xoff;
        } else if (!m.is_free && m.ob_table) {
            const double lSv = log_pos(PS_d);
            lw = fma(m.beta, log_pos(PL_d) - lSv, lSv);
        } else {
            lw = log_pos(prod1 * PS_d) +
                 m.beta * log_pos(fast_div(PL_d, PS_d));
        }
        constexpr bool LWP = QMC_LW_PAIRS && sizeof(R) == 8 && !ZW;
        const bool merged = LWP && !m.is_free && m.ob_table;
        if (sizeof(R) == 4)
            lw += LN2 * ((double)eS +
                         m.beta * (double)(ZW ? eL : eL - eS));
        if (!merged) {
            lw -= xoff;
            if (PAD && !live) lw = 0.0;
        }
    }
    if (WF && EN) {
        wave_sum2_mfma(e_lane, lw, E, logwf);
        E += e_consts;
    } else if (EN) {
        E = wave_sum_mfma(e_lane) + e_consts;
    } else {
        logwf = wave_sum_mfma(lw);
    }
}
