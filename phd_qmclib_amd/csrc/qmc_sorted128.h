// qmc_sorted128.h -- the pair sum on exactly ascending rows (qmc_sorted64.h) for
// the shape of BASELINE configs[3]: one walker per wavefront, TWO particles per
// lane, N = 128 (the multi-GPU headline workload).
//
// A lane holds the consecutive slots 2 gl, 2 gl + 1 of the sorted row.  At
// rotation step k it meets the two particles of lane gl - k: four pairs, at
// slot distances 2k - 1, 2k, 2k, 2k + 1.  The LDS tables are doubled as for one
// particle per lane (upper copy = the particle, lower copy = the particle one
// period below), so every pair is seen unwrapped and ordered, D' >= 0, and D'
// grows with the slot distance.  Hence:
//   * leading steps: ONE compare per step -- the farthest of the four pairs
//     (own slot 1 against the partner's slot 0) is short for every lane -- and
//     four one-case short-range pairs with no classification at all (the old
//     two-particle path had no such phase: every pair was classified and went
//     through the two-case branches);
//   * general steps: a pair is short iff |sin(pi D' / L)| < sin(pi rm / L), in
//     the single case; no generic branch;
//   * once per walker: the row is ascending and the farthest partner of the
//     rotation (slot distance 65) is closer than L - rm.
// Two steps per trip with two register sets for the requested partner tables.
#pragma once

#include "qmc_sorted64.h"

// the two table entries of a lane (slots 2 l, 2 l + 1) in one LDS access:
// ds_read_b128 in double (the rows keep the pairs 16-byte aligned,
// SortedRows<128>), ds_read_b64 in float
#ifndef QMC_S128_SU_PAIRS
#define QMC_S128_SU_PAIRS 1
#endif
template <typename R> struct SlotPair;
template <> struct SlotPair<double> {
    typedef double type __attribute__((ext_vector_type(2)));
};
template <> struct SlotPair<float> {
    typedef float type __attribute__((ext_vector_type(2)));
};
template <typename R>
__device__ __forceinline__ typename SlotPair<R>::type ld_pair(const R *p)
{
    return *(const typename SlotPair<R>::type *)p;
}

// ---- exact order of a row of 128 slots, two per lane ------------------------
// ascending: z0 <= z1 inside every lane and z1 <= the next lane's z0
__device__ __forceinline__ bool rows_ascending128(const double (&z)[2],
                                                  int nl = 64)
{
    // lane i takes the next lane's first slot (the last lane 0.0, unused:
    // bound_ctrl saves the copy of the old value)
    const int lo = __double2loint(z[0]), hi = __double2hiint(z[0]);
    const double up = __hiloint2double(
        __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true),
        __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true));
    const int lane = threadIdx.x & 63;
    const bool bad = (lane < nl) &
                     ((z[1] < z[0]) | ((lane < nl - 1) & (up < z[1])));
    return __builtin_amdgcn_ballot_w64(bad) == 0ull;
}

// (nl: lanes in use, each holding two particles)
__device__ __forceinline__ bool sort_rows128(double (&z)[2], int (&lab)[2], int gl,
                                             int nl = 64)
{
    if (rows_ascending128(z, nl)) return true;
    const int lane = threadIdx.x & 63;
    const int a_up = (lane >= nl - 1 ? lane : lane + 1) << 2;
    const int a_dn = (lane == 0 || lane >= nl ? lane : lane - 1) << 2;
    for (int it = 0; it < 66; ++it) {
        anchor_seam_rows<2>(z, lab, 2 * nl);
        // even phase: the two slots of a lane
        {
            const bool sw = lane < nl && z[1] < z[0];
            const double t = z[0]; const int u = lab[0];
            z[0] = sw ? z[1] : z[0]; lab[0] = sw ? lab[1] : lab[0];
            z[1] = sw ? t : z[1];    lab[1] = sw ? u : lab[1];
        }
        // odd phase: (slot 1 of a lane, slot 0 of the next); both lanes of a
        // pair see the same two values
        {
            const double up = __hiloint2double(
                __builtin_amdgcn_ds_bpermute(a_up, __double2hiint(z[0])),
                __builtin_amdgcn_ds_bpermute(a_up, __double2loint(z[0])));
            const int lup = __builtin_amdgcn_ds_bpermute(a_up, lab[0]);
            const double dn = __hiloint2double(
                __builtin_amdgcn_ds_bpermute(a_dn, __double2hiint(z[1])),
                __builtin_amdgcn_ds_bpermute(a_dn, __double2loint(z[1])));
            const int ldn = __builtin_amdgcn_ds_bpermute(a_dn, lab[1]);
            const bool t_up = lane < nl - 1 && up < z[1];
            const bool t_dn = lane > 0 && lane < nl && dn > z[0];
            z[1] = t_up ? up : z[1]; lab[1] = t_up ? lup : lab[1];
            z[0] = t_dn ? dn : z[0]; lab[0] = t_dn ? ldn : lab[0];
        }
        if (rows_ascending128(z, nl)) return true;
    }
    return false;
}

// the farthest pair of the rotation: own slot 1 against slot 0 of lane gl ^ 32
// (slot distance 65), closer than L - rm
__device__ __forceinline__ bool far_partner_ok128(const DevModel &m,
                                                  const double (&z)[2], int gl)
{
    const double zp = __shfl_xor(z[0], 32, 64);
    const double d = (gl < 32) ? (z[1] - zp) + m.L : z[1] - zp;
    return __builtin_amdgcn_ballot_w64(d >= m.L_minus_rm) == 0ull;
}

// the same for a ring of nl < 64 lanes
__device__ __forceinline__ bool far_partner_ok_ring128(const DevModel &m,
                                                       const double (&z)[2],
                                                       int gl, int nl)
{
    const bool live = gl < nl;
    int src = gl - nl / 2;
    const bool wrapped = src < 0;
    if (wrapped) src += nl;
    const double zp = __shfl(z[0], live ? src : gl, 64);
    const double d = wrapped ? (z[1] - zp) + m.L : z[1] - zp;
    return __builtin_amdgcn_ballot_w64(live & (d >= m.L_minus_rm)) == 0ull;
}

// One walker on an ascending row of 128 slots.  z[2]: the lane's particles
// (slots 2 gl, 2 gl + 1); lds: the 5 rows of sorted_particle_setup (194 entries
// each).
//   PAD : N < 128 particles, N even: the first nl = N / 2 lanes hold two each
//         (see eval_sorted64 for what changes on a ring shorter than the wave)
template <typename R, bool WF, bool EN, bool REUSE, bool PAD = false>
__device__ __forceinline__ void eval_sorted128(const DevModel &m,
                                               const double (&z)[2], int gl,
                                               double *lds, double (&F)[2],
                                               double &E, double &logwf)
{
    constexpr int G = 64, NS = 128, H = SortedRows<NS>::H,
                  ROW = SortedRows<NS>::ROW;
    constexpr bool COT = SortedCot<WF, EN, REUSE>::ON;   // (qmc_sorted64.h)
    constexpr bool TAN = SortedCot<WF, EN, REUSE>::TAN;
    typedef SortedCot<WF, EN, REUSE> RowsOf;
    R *lS = (R *)lds, *lC = lS + ROW, *lSU = lS + RowsOf::ROW_SU * ROW,
      *lCU = lS + 3 * ROW, *lZ = lS + RowsOf::ROW_Z * ROW;
    const int n = PAD ? m.n : NS;            // particles (even)
    const int nl = n / 2;                    // lanes in use
    const int K = nl / 2;                    // rotation steps
    const bool half_last = !PAD || (nl & 1) == 0;   // step K is a half step
    const int kfull = half_last ? K - 1 : K;
    const bool live = !PAD || gl < nl;
    const unsigned long long live_mask =
        PAD ? __builtin_amdgcn_ballot_w64(live) : ~0ull;
    [[maybe_unused]] constexpr int QMC_SEC_OFF = REUSE ? QMC_NSEC / 2 : 0;
    QMC_SECTION("tables+onebody");
    Own64<R> o[2];
    SortedOneBody ob[2];
    PTab ta[2];
    SortedPub pub[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
        sorted_particle_setup<R, WF, EN, REUSE, NS, false>(
            m, z[a], 2 * gl + a, (R *)lds, o[a], ob[a], n, ta[a], pub[a]);
    // both entries of the lane with one 16-byte store per row; the pair one
    // period below for the lanes the rotation reaches around the row's end
    // (COT: first row = cotangents, same one period below, no cosine row;
    // TAN: k2 sine row = kappa tan(k2 z), no k2 cosine row)
#define QMC_S128_ST(row, i, v0, v1)                                           \
        *(typename SlotPair<R>::type *)((row) + (i)) =                        \
            typename SlotPair<R>::type{ (R)(v0), (R)(v1) }
    if (!REUSE && live) {
        const int up = H + 2 * gl, lo = up - n;
        QMC_S128_ST(lS, up, pub[0].s, pub[1].s);
        if (!COT) QMC_S128_ST(lC, up, pub[0].c, pub[1].c);
        QMC_S128_ST(lSU, up, pub[0].su, pub[1].su);
        if (!TAN) QMC_S128_ST(lCU, up, pub[0].cu, pub[1].cu);
        QMC_S128_ST(lZ, up, z[0], z[1]);
        if (lo >= 2) {
            if (COT) {
                QMC_S128_ST(lS, lo, pub[0].s, pub[1].s);
            } else {
                QMC_S128_ST(lS, lo, -pub[0].s, -pub[1].s);
                QMC_S128_ST(lC, lo, -pub[0].c, -pub[1].c);
            }
            QMC_S128_ST(lSU, lo, pub[0].su_lo, pub[1].su_lo);
            if (!TAN) QMC_S128_ST(lCU, lo, pub[0].cu_lo, pub[1].cu_lo);
            QMC_S128_ST(lZ, lo, z[0] - m.L, z[1] - m.L);
        }
    } else if (REUSE && COT && live) {
        // the energy pass after an accepted VMC move: sine row -> cotangent
        // row, k2 sine row -> tangent row
        const int up = H + 2 * gl, lo = up - n;
        QMC_S128_ST(lS, up, pub[0].s, pub[1].s);
        if (TAN) QMC_S128_ST(lSU, up, pub[0].su, pub[1].su);
        if (lo >= 2) {
            QMC_S128_ST(lS, lo, pub[0].s, pub[1].s);
            if (TAN) QMC_S128_ST(lSU, lo, pub[0].su_lo, pub[1].su_lo);
        }
    }
#undef QMC_S128_ST
    if (!REUSE || COT) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    int nb_wave = 0;
    const bool nb_counted = EN && !m.is_free && m.ob_table && m.uniform_barrier;
    if (nb_counted)
        nb_wave = __popcll(__ballot(ob[0].barrier & live)) +
                  __popcll(__ballot(ob[1].barrier & live));
    const R sin_rm = (R)m.sin_rm;
    // particle b of the partner lane of step k: entry (H + 2 gl) - 2 k + b
    const R *pS = lS + H + 2 * gl, *pC = (COT ? lZ : lC) + H + 2 * gl,
            *pSU = lSU + H + 2 * gl, *pCU = lCU + H + 2 * gl,
            *pZ = lZ + H + 2 * gl;

    // the lane below in the ring of the lanes in use (travelling sums)
    int ring_src = 0;
    if (PAD) ring_src = (live ? (gl == 0 ? nl - 1 : gl - 1) : gl) << 2;
#define QMC_S128_ROR(T)                                                       \
    (PAD ? __hiloint2double(                                                  \
               __builtin_amdgcn_ds_bpermute(ring_src, __double2hiint((double)(T))), \
               __builtin_amdgcn_ds_bpermute(ring_src, __double2loint((double)(T)))) \
         : (double)group_ror1<G>(T))
#define QMC_S128_ALL(cond)                                                    \
    ((__builtin_amdgcn_ballot_w64(cond) | ~live_mask) == ~0ull)
    R Fr[2] = { 0, 0 };      // drift: pair quotients in units of a_long (Own64)
    // (one register pair across the loops instead of two)
    const double kin1_sum = ob[0].kin1 + ob[1].kin1;
    R T[2] = { 0, 0 };       // travelling sums for the partner lane's particles
    R Qall = 0, Qs = 0;      // sum of q^2 over all / short pairs
    R PS = 1, PL = 1;        // products: short factors f2, |Y| of all pairs
    int eS = 0, eL = 0;
    int ns = 0;              // short pairs (EN)

    // a short-range pair of own particle `oa` with the partner's k2-table
    // (bsu, bcu): numerator / denominator of the one-case form
    // (TAN, qmc_sorted64.h: bsu = the partner's kappa tan(k2 z'), bcu unused)
#define QMC_S128_SHORT_XY(oa, bsu, bcu, X, Y)                                 \
    const R Y = TAN ? q_fma((oa).c0, (bsu), (R)1)                             \
                    : (oa).c0 * (bcu) + (oa).s0 * (bsu);                      \
    R X = 0;                                                                  \
    if (EN) X = TAN ? (oa).s0 - (bsu)                                         \
                    : (oa).ks0 * (bcu) - (oa).kc0 * (bsu);
    // a pair of a general step: class from the sine, short ones recomputed in
    // an exec-masked region
    // (bsu, bcu: the partner's k2-table entry.  QMC_S128_SU_PAIRS: read for
    // both partner slots with one 16-byte access per row at the top of the
    // step -- a step of the general phase nearly always has short pairs in some
    // lane, an LDS instruction costs the same for one lane as for 64, and a
    // read inside the exec-masked region is a lane-stride-16 access (2-way
    // bank conflict) whose latency nothing hides; 0: read where it is used)
    // (COT, qmc_sorted64.h: cs = the partner's cotangent, cc = its position)
#define QMC_S128_XY(oa, cs, cc, bsu, bcu, X, Y, sh)                           \
    const R Y##_s = COT ? (cs) - (oa).s              /* t_j - t_i */           \
                        : (oa).s * (cc) - (oa).c * (cs);  /* sin(pi D'/L) >= 0 */ \
    R X = 0;                                                                  \
    if (EN) X = COT ? q_fma((oa).s, (cs), (R)1)      /* t_i t_j + 1 */         \
                    : (oa).c * (cc) + (oa).s * (cs); /* cos(pi D' / L) */      \
    const bool sh = COT ? (cc) > (oa).zt             /* D' < rm */             \
                        : q_abs(Y##_s) < sin_rm;                              \
    /* (the mask of the lanes with a short pair, taken where the compare is:  \
       asked for after the exec-masked region it is rebuilt from a select) */ \
    const unsigned long long sh##_m = __builtin_amdgcn_ballot_w64(sh);        \
    R Y = Y##_s;                                                              \
    if (sh) {                                                                 \
        asm volatile("");                                                     \
        const R bsu_ = (bsu);                                                 \
        if (TAN) {                                                            \
            Y = q_fma((oa).c0, bsu_, (R)1);                                   \
            X = (oa).s0 - bsu_;                                               \
        } else {                                                              \
            const R bcu_ = (bcu);                                             \
            Y = (oa).c0 * bcu_ + (oa).s0 * bsu_;                              \
            if (EN) X = (oa).ks0 * bcu_ - (oa).kc0 * bsu_;                    \
        }                                                                     \
    }

    // ---- k = 0: the pair inside the lane (slot 1 against slot 0) ----
    QMC_SECTION("pairs_in_lane");
    {
        // (COT: o.s is the cotangent and o.c the position, sorted_particle_setup)
        const R cs = (R)o[0].s, cc = (R)o[0].c;
        QMC_S128_XY(o[1], cs, cc, pSU[0], pCU[0], X, Y, sh)
        // (the products of an idle lane never reach the sums: its log is dropped)
        if (WF) { PL *= Y; if (sh) { asm volatile(""); PS *= Y; } }
        if (EN) {
            const R q = pair_div(X, Y);
            Fr[1] += q; Fr[0] -= q;
            Qall = q_fma(q, q, Qall);
            ns += __popcll(sh_m & live_mask);
            if (sh) { asm volatile(""); Qs = q_fma(q, q, Qs); }
        }
    }

    // ---- leading steps: all four pairs of every lane are short ----
    QMC_SECTION("leading_short_steps");
    int k = 1;
    R Pl = 1;                // factors of the leading steps
    R Ql = 0;
#define QMC_S128_LEAD(su0, cu0, su1, cu1)                                     \
    {                                                                         \
        QMC_S128_SHORT_XY(o[0], su0, cu0, X00, Y00)                           \
        QMC_S128_SHORT_XY(o[1], su0, cu0, X10, Y10)                           \
        QMC_S128_SHORT_XY(o[0], su1, cu1, X01, Y01)                           \
        QMC_S128_SHORT_XY(o[1], su1, cu1, X11, Y11)                           \
        if (WF) Pl *= (Y00 * Y10) * (Y01 * Y11);                              \
        if (EN) {                                                             \
            const R q00 = pair_div(X00, Y00), q10 = pair_div(X10, Y10);       \
            const R q01 = pair_div(X01, Y01), q11 = pair_div(X11, Y11);       \
            Fr[0] += q00 + q01;                                               \
            Fr[1] += q10 + q11;                                               \
            T[0] -= q00 + q10;                                                \
            T[1] -= q01 + q11;                                                \
            Ql = q_fma(q00, q00, Ql); Ql = q_fma(q10, q10, Ql);               \
            Ql = q_fma(q01, q01, Ql); Ql = q_fma(q11, q11, Ql);               \
            T[0] = (R)QMC_S128_ROR(T[0]);                                     \
            T[1] = (R)QMC_S128_ROR(T[1]);                                     \
        }                                                                     \
    }
    {
        // set A: step k, set B: step k + 1 (entries of both partner particles)
        typedef typename SlotPair<R>::type R2;
        // (TAN: the cosine row of the k2-table is neither written nor read)
        R2 asu = ld_pair(pSU - 2), acu = TAN ? asu : ld_pair(pCU - 2);
        R az = pZ[-2];
        R2 bsu = ld_pair(pSU - 4), bcu = TAN ? bsu : ld_pair(pCU - 4);
        R bz = pZ[-4];
        // (the farthest pair of a step: own slot 1 against the partner's slot 0)
        const R zt = o[1].zt;
#pragma clang loop unroll(disable)
        while (k < kfull) {
            if (!QMC_S128_ALL(az > zt)) break;
            QMC_S128_LEAD(asu[0], acu[0], asu[1], acu[1])
            asu = ld_pair(pSU - 2 * (k + 2));
            if (!TAN) acu = ld_pair(pCU - 2 * (k + 2));
            az = pZ[-2 * (k + 2)];
            ++k;
            if (!QMC_S128_ALL(bz > zt)) break;
            QMC_S128_LEAD(bsu[0], bcu[0], bsu[1], bcu[1])
            bsu = ld_pair(pSU - 2 * (k + 2));
            if (!TAN) bcu = ld_pair(pCU - 2 * (k + 2));
            bz = pZ[-2 * (k + 2)];
            ++k;
            if (WF) {
                // (eight factors per trip)
                int e = 0;
                q_fold(Pl, e);
                eS += e; eL += e;
            }
        }
    }
#undef QMC_S128_LEAD
    // (these pairs belong to both products and both sums)
    if (WF) { PS *= Pl; PL *= Pl; }
    if (EN) { Qall += Ql; Qs += Ql; ns += 4 * (k - 1) * nl; }

    // ---- general steps ----
    QMC_SECTION("rotation_loop_body");
    // the four pairs of a step against partner tables (s0, c0), (s1, c1)
#if QMC_S128_SU_PAIRS
#define QMC_S128_SU_LOAD(kk)                                                  \
        const typename SlotPair<R>::type su_ = ld_pair(pSU - 2 * (kk)),       \
            cu_ = TAN ? su_ : ld_pair(pCU - 2 * (kk));
#define QMC_S128_SU(kk, b) su_[b]
#define QMC_S128_CU(kk, b) cu_[b]
#else
#define QMC_S128_SU_LOAD(kk)
#define QMC_S128_SU(kk, b) pSU[-2 * (kk) + (b)]
#define QMC_S128_CU(kk, b) pCU[-2 * (kk) + (b)]
#endif
#define QMC_S128_STEP(s0_, c0_, s1_, c1_, kk, LAST)                           \
    {                                                                         \
        const bool mine = live & (!(LAST) || gl < K);                         \
        QMC_S128_SU_LOAD(kk)                                                  \
        QMC_S128_XY(o[0], s0_, c0_, QMC_S128_SU(kk, 0), QMC_S128_CU(kk, 0), X00, Y00, h00) \
        QMC_S128_XY(o[1], s0_, c0_, QMC_S128_SU(kk, 0), QMC_S128_CU(kk, 0), X10, Y10, h10) \
        QMC_S128_XY(o[0], s1_, c1_, QMC_S128_SU(kk, 1), QMC_S128_CU(kk, 1), X01, Y01, h01) \
        QMC_S128_XY(o[1], s1_, c1_, QMC_S128_SU(kk, 1), QMC_S128_CU(kk, 1), X11, Y11, h11) \
        if (WF && mine) {                                                     \
            PL *= (Y00 * Y10) * (Y01 * Y11);                                  \
            if (h00) { asm volatile(""); PS *= Y00; }                         \
            if (h10) { asm volatile(""); PS *= Y10; }                         \
            if (h01) { asm volatile(""); PS *= Y01; }                         \
            if (h11) { asm volatile(""); PS *= Y11; }                         \
        }                                                                     \
        if (EN) {                                                             \
            const R q00 = pair_div(X00, Y00), q10 = pair_div(X10, Y10);       \
            const R q01 = pair_div(X01, Y01), q11 = pair_div(X11, Y11);       \
            Fr[0] += q00 + q01;                                               \
            Fr[1] += q10 + q11;                                               \
            if (!(LAST)) {                                                    \
                T[0] -= q00 + q10;                                            \
                T[1] -= q01 + q11;                                            \
                T[0] = (R)QMC_S128_ROR(T[0]);                                 \
                T[1] = (R)QMC_S128_ROR(T[1]);                                 \
            }                                                                 \
            const unsigned long long mine_m =                                 \
                (LAST) ? __builtin_amdgcn_ballot_w64(mine) : live_mask;       \
            ns += __popcll(h00_m & mine_m) + __popcll(h10_m & mine_m) +       \
                  __popcll(h01_m & mine_m) + __popcll(h11_m & mine_m);        \
            if (mine) {                                                       \
                Qall = q_fma(q00, q00, Qall); Qall = q_fma(q10, q10, Qall);   \
                Qall = q_fma(q01, q01, Qall); Qall = q_fma(q11, q11, Qall);   \
                if (h00) { asm volatile(""); Qs = q_fma(q00, q00, Qs); }      \
                if (h10) { asm volatile(""); Qs = q_fma(q10, q10, Qs); }      \
                if (h01) { asm volatile(""); Qs = q_fma(q01, q01, Qs); }      \
                if (h11) { asm volatile(""); Qs = q_fma(q11, q11, Qs); }      \
            }                                                                 \
        }                                                                     \
    }
    // ... and of a step whose four pairs are long-range for every lane: no
    // classification, no exec-masked region
#define QMC_S128_LONG_XY(oa, cs, cc, X, Y)                                    \
    const R Y = COT ? (cs) - (oa).s : (oa).s * (cc) - (oa).c * (cs);          \
    R X = 0;                                                                  \
    if (EN) X = COT ? q_fma((oa).s, (cs), (R)1)                               \
                    : (oa).c * (cc) + (oa).s * (cs);
#define QMC_S128_LONG_STEP(s0_, c0_, s1_, c1_)                                \
    {                                                                         \
        QMC_S128_LONG_XY(o[0], s0_, c0_, X00, Y00)                            \
        QMC_S128_LONG_XY(o[1], s0_, c0_, X10, Y10)                            \
        QMC_S128_LONG_XY(o[0], s1_, c1_, X01, Y01)                            \
        QMC_S128_LONG_XY(o[1], s1_, c1_, X11, Y11)                            \
        if (WF && live) PL *= (Y00 * Y10) * (Y01 * Y11);                      \
        if (EN) {                                                             \
            const R q00 = pair_div(X00, Y00), q10 = pair_div(X10, Y10);       \
            const R q01 = pair_div(X01, Y01), q11 = pair_div(X11, Y11);       \
            Fr[0] += q00 + q01;                                               \
            Fr[1] += q10 + q11;                                               \
            T[0] -= q00 + q10;                                                \
            T[1] -= q01 + q11;                                                \
            T[0] = (R)QMC_S128_ROR(T[0]);                                     \
            T[1] = (R)QMC_S128_ROR(T[1]);                                     \
            Qall = q_fma(q00, q00, Qall); Qall = q_fma(q10, q10, Qall);       \
            Qall = q_fma(q01, q01, Qall); Qall = q_fma(q11, q11, Qall);       \
        }                                                                     \
    }
    {
        typedef typename SlotPair<R>::type R2;
        R2 as = ld_pair(pS - 2 * k), ac = ld_pair(pC - 2 * k);
        R2 bs = ld_pair(pS - 2 * (k + 1)), bc = ld_pair(pC - 2 * (k + 1));
        // the nearest pair of a step: own slot 0 against the partner's slot 1;
        // once it is long-range for every lane, so is every later pair
        const R zt0 = o[0].zt;
#pragma clang loop unroll(disable)
        while (k < kfull) {
            if (__builtin_amdgcn_ballot_w64(live & (pZ[-2 * k + 1] > zt0)) ==
                0ull)
                break;
            QMC_S128_STEP(as[0], ac[0], as[1], ac[1], k, false)
            as = ld_pair(pS - 2 * (k + 2)); ac = ld_pair(pC - 2 * (k + 2));
            QMC_S128_STEP(bs[0], bc[0], bs[1], bc[1], k + 1, false)
            // (k + 3 <= N / 4 + 2: the two unused entries in front of the rows)
            bs = ld_pair(pS - 2 * (k + 3)); bc = ld_pair(pC - 2 * (k + 3));
            k += 2;
            if (WF) {
                q_fold(PS, eS);
                q_fold(PL, eL);
            }
        }
        // ---- trailing steps: every pair is long-range ----
        QMC_SECTION("rotation");
#pragma clang loop unroll(disable)
        while (k < kfull) {
            QMC_S128_LONG_STEP(as[0], ac[0], as[1], ac[1])
            as = ld_pair(pS - 2 * (k + 2)); ac = ld_pair(pC - 2 * (k + 2));
            QMC_S128_LONG_STEP(bs[0], bc[0], bs[1], bc[1])
            bs = ld_pair(pS - 2 * (k + 3)); bc = ld_pair(pC - 2 * (k + 3));
            k += 2;
            if (WF) q_fold(PL, eL);
        }
        if (k <= kfull) {
            QMC_S128_STEP(as[0], ac[0], as[1], ac[1], k, false)
            ++k;
            as = bs; ac = bc;
        }
        // the final half step (an even number of lanes in use) visits every
        // pair from both sides
        QMC_SECTION("rotation_last_step");
        if (half_last) QMC_S128_STEP(as[0], ac[0], as[1], ac[1], k, true)
    }
#undef QMC_S128_STEP
#undef QMC_S128_SU_LOAD
#undef QMC_S128_SU
#undef QMC_S128_CU
#undef QMC_S128_LONG_STEP
#undef QMC_S128_LONG_XY
#undef QMC_S128_XY
#undef QMC_S128_SHORT_XY
    if (EN) {
        // after kfull rotations lane l holds the sums of lane l - kfull - 1
        if (PAD) {
            int src = gl + kfull + 1;
            if (src >= nl) src -= nl;
            src = live ? src : gl;
            Fr[0] += __shfl(T[0], src, 64);
            Fr[1] += __shfl(T[1], src, 64);
        } else {
            Fr[0] += __shfl_xor(T[0], G / 2, 64);
            Fr[1] += __shfl_xor(T[1], G / 2, 64);
        }
    }
#undef QMC_S128_ROR
#undef QMC_S128_ALL

    QMC_SECTION("energy+logwf");
    double e_lane = 0.0, e_consts = 0.0;
    if (EN) {
        F[0] = fma(m.a_long, (double)Fr[0], ob[0].ldz);
        F[1] = fma(m.a_long, (double)Fr[1], ob[1].ldz);
        const double Qall_d = (double)Qall, Qs_d = (double)Qs;
        const double pk = (Qs_d + (Qall_d - Qs_d) * m.inv_beta) * m.a_long_sq;
        e_lane = fma(2.0, pk, kin1_sum) - F[0] * F[0] - F[1] * F[1];
        if (PAD && !live) e_lane = 0.0;
        if (nb_counted)
            e_consts += (double)(n - nb_wave) * m.e0 +
                        (double)nb_wave * (m.v_barrier - m.v0_minus_e0);
        const int nlong = n * (n - 1) / 2 - ns;
        e_consts += 2.0 * (m.k2sq * (double)ns + m.b_long * (double)nlong);
    }
    double lw = 0.0;
    if (WF) {
        const double LN2 = 0.693147180559945309417;
        q_fold(PS, eS);
        q_fold(PL, eL);
        const double PS_d = (double)PS, PL_d = (double)PL;
        if (!m.is_free && m.ob_table) {
            const double lSv = log_pos(PS_d);
            lw = fma(m.beta, log_pos(PL_d) - lSv, lSv);
        } else {
            lw = log_pos(ob[0].prod1 * ob[1].prod1 * PS_d) +
                 m.beta * log_pos(fast_div(PL_d, PS_d));
        }
        lw += LN2 * ((double)eS + m.beta * (double)(eL - eS));
        lw -= ob[0].xoff + ob[1].xoff;
        if (PAD && !live) lw = 0.0;
    }
    // (two particles per lane: the sums over the lanes by DPP rotations and a
    // butterfly, as eval_walker does for this shape -- the accumulators of
    // the matrix instruction cost it a wave of occupancy)
    if (EN) E = group_sum<G>(e_lane) + e_consts;
    if (WF) logwf = group_sum<G>(lw);
}
