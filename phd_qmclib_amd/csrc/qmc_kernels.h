// qmc_kernels.h -- the __global__ kernels of libqmcwalk.so and their argument
// blocks (host API and launch code: qmcwalk.hip; device building blocks:
// qmc_device.h).  Kept in a header so that a single instantiation can be
// compiled on its own (tools/isa_one.sh: ISA dumps, instruction counts).
#pragma once

#include "qmc_device.h"
#include "qmc_sorted64.h"
#include "qmc_sorted128.h"

#ifndef QMC_SORTED128
#define QMC_SORTED128 1
#endif

static constexpr int BLOCK = 256;          // 4 wavefronts per workgroup

// Workgroup size of the walker kernels.  One walker per wavefront (G = 64):
// ONE wavefront per workgroup.  A workgroup's slots are released when its last
// wavefront ends, and the wavefronts of the VMC step do not take equally long
// (an accepted move runs the energy pass, a rejected one does not): with four
// per workgroup 1 - 0.52^4 = 93 % of the workgroups last as long as an accepted
// move and the two-pass step gains nothing (measured: 10 % fewer busy vector
// cycles, 5 % MORE time); with one, a slot is free again as soon as its chain is
// done.  The kernels synchronise inside a wavefront only, so the size is free.
#ifndef QMC_BLOCK64
#define QMC_BLOCK64 64
#endif
template <int G>
struct WalkBlock {
    static constexpr int N = (G == 64) ? QMC_BLOCK64 : BLOCK;
};

// Which walker a workgroup of the one-wavefront-per-workgroup kernels owns.
// Workgroups are handed to the 8 XCDs round-robin (workgroup b runs on XCD
// b mod 8) and every XCD has its own L2.  The per-walker scalars (log|psi|,
// carried energy, block sums; DMC: energy, weight, cloning reference) are 8
// bytes each, 16 walkers to a 128-byte line: with walker = workgroup index the
// 16 walkers of a line sit on 8 different XCDs and every L2 fetches and writes
// back the whole line for its two walkers (measured: 1699 bytes of HBM traffic
// per VMC chain-step against the algorithmic 1056).  Here the 16 walkers of a
// line go to ONE XCD: within every run of 128 workgroups, XCD x takes walkers
// 16 x .. 16 x + 15.  (The launch grid is rounded up to a multiple of 128.)
#ifndef QMC_XCD_MAP
#define QMC_XCD_MAP 1
#endif
template <int GPB>
__device__ __forceinline__ long long walker_of_block(unsigned b, int grp)
{
    if (QMC_XCD_MAP && GPB == 1)
        return ((long long)(b >> 7) << 7) + ((b & 7u) << 4) + ((b & 127u) >> 3);
    return (long long)b * GPB + grp;
}

// LDS doubles per lane group of the stepping kernels (vmc_step, dmc_evolve):
// the sorted-row path of the exact N = 128 shape keeps the positions as well,
// 5 rows of 192 entries (qmc_sorted64.h: sorted_particle_setup)
// (DMC = the energy-only stepping kernel: three rows with the cotangent /
// tangent tables, qmc_sorted64.h: SortedCot::COMPACT)
template <int G, int P, bool PAD, bool ZC, bool DMC = false>
struct StepLds {
    static constexpr int NROWS =
        DMC ? SortedCot<false, true, false>::ROWS : 5;
    // (+ one row of partner sums, qmc_sorted64.h: QMC_T_LDS)
    static constexpr int SORTED =
        (QMC_SORTED128 && G == 64 && P == 2 && !ZC)
            ? NROWS * SortedRows<128>::ROW
            : (QMC_SORTED64 && G == 64 && P == 1 && !ZC)
                  ? (NROWS + (QMC_T_LDS ? 1 : 0)) * SortedRows<64>::ROW
                  : 0;
    // a walker that fails the per-walker checks of the sorted-row path runs
    // eval_walker on the same LDS region: room for the larger of the two
    // layouts (N <= 64: 485 doubles sorted, 512 general -- ADVICE r3)
    static constexpr int GENERAL = GroupLds<G, P, ZC>::DOUBLES;
    static constexpr int DOUBLES = SORTED > GENERAL ? SORTED : GENERAL;
};

// a wave-uniform 64-bit value as the compiler can see it (scalar registers)
__device__ __forceinline__ long long qmc_uniform(long long v)
{
    const int lo = __builtin_amdgcn_readfirstlane((int)v);
    const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return ((long long)hi << 32) | (unsigned int)lo;
}

// Minimum waves per SIMD asked of the register allocator for the N <= 64
// shape.  Round 2 held it to 64 registers (8 waves, one spilled double: +1.5 %
// then).  With the sorted-lane path the kernel needs 70 left to itself (7
// waves, no scratch) and 64 + 20 bytes of scratch at 8 waves; the two measure
// the same (VMC 1.897 / 1.898 ms, DMC 0.522 / 0.519 ms, profiles/
// r03_ab_variants.txt): the build without scratch ships.
// (The N <= 128 shape held to 80 registers for 6 waves spills 56-140 bytes per
// lane and loses 10-15 %.)
#ifndef QMC_LB_P1
#define QMC_LB_P1 7
#endif
// (the sorted-row N = 128 shape: 7.5 KB of LDS per walker leaves room for five
// wavefronts per SIMD, which takes <= 96 registers)
#ifndef QMC_LB_DMC_P2
#define QMC_LB_DMC_P2 5
#endif
#define QMC_LB_WAVES , ((G == 64 && P == 1) ? QMC_LB_P1 : 1)
#define QMC_LB_WAVES_DMC , ((G == 64 && P == 1) ? QMC_LB_P1 \
                            : (G == 64 && P == 2 && !ZC) ? QMC_LB_DMC_P2 : 1)
// (The VMC step of the exact N <= 128 shape held to 96 registers for a fifth
// wave was 2.4 % faster with the four-case form; with the two-case form it needs
// 108 registers unconstrained (4 waves) and the constraint costs 2.5 %: off.)
// (Round 4: the production instantiation needs 94 on its own -- five waves --
// but the series / tape instantiation was allotted 129, three waves: held to
// four.)
#ifndef QMC_LB_VMC_P2
#define QMC_LB_VMC_P2 4
#endif
#define QMC_LB_WAVES_VMC , ((G == 64 && P == 1) ? QMC_LB_P1 \
                            : (G == 64 && P == 2 && !ZC) ? QMC_LB_VMC_P2 : 1)

// The VMC step in two passes (one wavefront per walker: the accept decision
// is wave-uniform): log|psi| of the proposal first -- no quotient, no drift --
// then, for accepted moves only, energy and drift from the pair tables still in
// LDS.  The reference computes the energy of accepted moves only as well
// (qmc_base/jastrow/vmc.py:253-262); a rejected proposal (53 % at the benchmark's
// move spread) pays for log|psi| alone.
#ifndef QMC_VMC_TWO_PASS
#define QMC_VMC_TWO_PASS 1
#endif

// The second normal of a Box-Muller pair is kept for the next time step of the
// slot (1) or computed again there (0: same stream, no `spare` array -- 8 N
// bytes less to write on even and to read on odd steps, a Philox block, a
// logarithm, a square root and a sincos per particle more on odd ones;
// VERDICT r3 item 5 asked for the A/B: profiles/r04_ab_variants.txt, 10).
// Measured (same box): without the cache the N = 64 step is 2.5 % slower
// (0.4775 against 0.4658 ms; 865 against 800 vector instructions per
// walker-step) with 2374 instead of 2885 bytes of HBM traffic per walker-step
// (1.13x the algorithmic bytes instead of 1.37x); at N = 128, where the
// pair loop is four times the per-particle work, the two tie (1.272 / 1.279
// ms).  The kernel is nowhere near the HBM roofline, so one particle per lane
// keeps the cache; two or more per lane do without it -- 1 KiB less traffic
// per walker-step at N = 128 and no 4.3 GB array at 2^22 walkers.
#ifndef QMC_DMC_SPARE
#define QMC_DMC_SPARE 1
#endif
template <int P>
struct DmcSpare {
    static constexpr bool ON = QMC_DMC_SPARE && P == 1;
};

// odd-even transposition passes (resort_step) run every this many steps
#ifndef QMC_RESORT_EVERY
#define QMC_RESORT_EVERY 4
#endif

// --------------------------------------------------------------- kernels ---
struct EvalArgs {
    const double *pos;   // [W][N]
    double *wf, *energy; // [W]
    double *ith, *drift; // [W][N]
    long long nconf;
};

template <int G, int P, bool PAD, bool ZC, typename R = double>
__global__ void __launch_bounds__(WalkBlock<G>::N QMC_LB_WAVES)
evaluate_kernel(const DevModel *__restrict__ mp, EvalArgs a)
{
    // model constants live in device memory: scalar loads on demand keep the
    // SGPR file free for the hot loop (by-value they overflow it)
    const DevModel &m = *mp;
    [[maybe_unused]] constexpr int QMC_SEC_OFF = 0;
    QMC_SECTION("top");
    extern __shared__ double smem[];
    constexpr int GPB = WalkBlock<G>::N / G;            // groups per block
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    double *lds = smem + (size_t)grp * GroupLds<G, P, ZC>::DOUBLES;
    const long long w = walker_of_block<GPB>(blockIdx.x, grp);
    const bool active = w < a.nconf;
    if (GPB == 1 && !active) return;      // (no one else in the workgroup)
    const long long wr = active ? w : 0;
    double z[P], z1[P], F[P], ei[P], E, wf;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int i = lane_particle<G, P, PAD>(m, gl, p);
        // (any position is accepted, as by the reference's functions: pair
        // tables from the image inside the box, one-body factor from the
        // position as given -- eval_walker)
        z1[p] = (i < m.n) ? a.pos[wr * m.n + i] : 0.0;
        z[p] = wrap_box(z1[p], m.L);
    }
    eval_walker<G, P, PAD, true, true, ZC, R>(m, z, z1, gl, lds, F, ei, E, wf);
    if (!active) return;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int i = lane_particle<G, P, PAD>(m, gl, p);
        if (i < m.n) {
            if (a.ith) a.ith[w * m.n + i] = ei[p];
            if (a.drift) a.drift[w * m.n + i] = F[p];
        }
    }
    if (gl == 0) {
        if (a.wf) a.wf[w] = wf;
        if (a.energy) a.energy[w] = E;
    }
}

// Energy + drift only (no log|psi|, no per-particle energies): the DMC
// build_state pass (qmc_base/jastrow/dmc.py:1043-1078).
struct PrepArgs {
    const double *pos;
    double *drift, *energy;
    long long nconf;
};

template <int G, int P, bool PAD, bool ZC, typename R = double>
__global__ void __launch_bounds__(WalkBlock<G>::N QMC_LB_WAVES)
prepare_kernel(const DevModel *__restrict__ mp, PrepArgs a)
{
    // model constants live in device memory: scalar loads on demand keep the
    // SGPR file free for the hot loop (by-value they overflow it)
    const DevModel &m = *mp;
    [[maybe_unused]] constexpr int QMC_SEC_OFF = 0;
    QMC_SECTION("top");
    extern __shared__ double smem[];
    constexpr int GPB = WalkBlock<G>::N / G;
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    double *lds = smem + (size_t)grp * GroupLds<G, P, ZC>::DOUBLES;
    const long long w = walker_of_block<GPB>(blockIdx.x, grp);
    const bool active = w < a.nconf;
    if (GPB == 1 && !active) return;      // (no one else in the workgroup)
    const long long wr = active ? w : 0;
    double z[P], z1[P], F[P], ei[P], E, wf;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int i = lane_particle<G, P, PAD>(m, gl, p);
        z1[p] = (i < m.n) ? a.pos[wr * m.n + i] : 0.0;
        z[p] = wrap_box(z1[p], m.L);
    }
    eval_walker<G, P, PAD, false, false, ZC, R>(m, z, z1, gl, lds, F, ei, E,
                                                wf);
    if (!active) return;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int i = lane_particle<G, P, PAD>(m, gl, p);
        if (i < m.n) a.drift[w * m.n + i] = F[p];
    }
    if (gl == 0) a.energy[w] = E;
}

// ---- VMC: one launch = one generator yield of every chain ----------------
// (a step loop inside the kernel lets LICM hoist ~40 polynomial constants and
// the model constants across it, tripling the register count; with the loop on
// the host the kernel has the register footprint of `evaluate_kernel` and the
// state round trip is ~1 KB per chain-step, far below the HBM roofline.)
struct VmcArgs {
    double *pos;          // [W][N] in/out, lane (position) order
    unsigned short *label;// [W][N] in/out, original index of each lane's particle
    double *wf;           // [W]    in/out  log|psi|
    double *ecarry;       // [W]    in/out  energy carried to rejected moves
    double *sum_e, *sum_e2;   // [W] running block sums
    long long *n_acc;
    double *ser_wf, *ser_e;   // [nyield][W] or null
    unsigned char *ser_stat;
    double *ser_pos;          // [nyield][W][N] or null
    const double *tape;   // [W][tape_steps][N+1] or null
    long long tape_steps;
    long long tape_idx;   // real step index into the tape for this yield
    long long W;
    long long y;          // yield index inside the block (series row)
    int forced;           // this yield is the initial state (ACCEPTED)
    int reset_sums;       // first yield of a block: sums start from zero
    int gaussian;
    unsigned int step;    // Philox step counter of this yield
    unsigned int chain0;
    unsigned long long seed;
    double move_spread;
};

// LEAN = the production path (Philox uniform proposal, per-chain block sums
// only); the full variant adds the test-only tape replay, the Gaussian
// proposal and the per-step series.
template <int G, int P, bool PAD, bool ZC, bool LEAN, typename R = double>
__global__ void __launch_bounds__(WalkBlock<G>::N QMC_LB_WAVES_VMC)
vmc_step_kernel(const DevModel *__restrict__ mp, VmcArgs a)
{
    const DevModel &m = *mp;
    [[maybe_unused]] constexpr int QMC_SEC_OFF = 0;
    QMC_SECTION("top");
    extern __shared__ double smem[];
    constexpr int GPB = WalkBlock<G>::N / G;
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    static_assert(StepLds<G, P, PAD, ZC>::DOUBLES >= GroupLds<G, P, ZC>::DOUBLES,
                  "eval_walker is the fallback on the same LDS region");
    double *lds = smem + (size_t)grp * StepLds<G, P, PAD, ZC>::DOUBLES;
    const long long w = walker_of_block<GPB>(blockIdx.x, grp);
    const bool active = w < a.W;
    if (GPB == 1 && !active) return;      // (no one else in the workgroup)
    const long long wr = active ? w : 0;
    const int n = m.n;
    const unsigned int slot = a.chain0 + (unsigned int)wr;
    // The very first yield of a generator is the initial state itself,
    // flagged ACCEPTED (qmc_base/vmc.py:616-618): a forced zero move.
    const bool forced = a.forced != 0;

    // The chain's scalars are needed only after the pair sum; their loads are
    // issued here so that the memory latency runs under it.  One chain per
    // wavefront: the index is wave-uniform and the loads are scalar (SGPR
    // results, no vector registers held across the pair sum).
    const long long wl = (G == 64) ? qmc_uniform(wr) : wr;
    double wf_cur = a.wf[wl];
    double e_cur = a.ecarry[wl];
    double se = 0.0, se2 = 0.0;
    long long na = 0;
    if (G == 64 && !a.reset_sums) {
        se = a.sum_e[wl]; se2 = a.sum_e2[wl]; na = a.n_acc[wl];
    }

    QMC_SECTION("load+philox+wrap");
    double zn[P];
    int labn[P];              // original particle index held by each lane
    double ua = 1.0;          // accept uniform
    // second words of the move blocks of particles 0 and 1 (the accept draw),
    // in the lanes that hold those particles
    unsigned int aw0 = 0u, aw1 = 0u;
    bool has0 = false, has1 = false;
    bool outside = false;     // forced yield: a particle given outside [0, L)
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int i = lane_particle<G, P, PAD>(m, gl, p);
        const double zp = (i < n) ? a.pos[wr * n + i] : 0.0;
        labn[p] = (i < n) ? (int)a.label[wr * n + i] : i;
        const unsigned li = (unsigned)labn[p];
        double d = 0.0;
        if (!forced && i < n) {
            if (!LEAN && a.tape) {
                double tv = a.tape[(wr * a.tape_steps + a.tape_idx) * (n + 1) + li];
                d = a.gaussian ? a.move_spread * tv
                               : (tv - 0.5) * a.move_spread;
            } else if (!LEAN && a.gaussian) {
                double g0, g1;
                philox_normal2(a.seed, slot, a.step, li, STREAM_VMC_MOVE, g0,
                               g1);
                d = a.move_spread * g0;
            } else {
                uint32_t w0, w1;
                vmc_move_block(a.seed, slot, a.step, li, w0, w1);
                d = vmc_move_unit(w0) * a.move_spread;
                if (li == 0u) { aw0 = w1; has0 = true; }
                if (li == 1u || n == 1) { aw1 = w1; has1 = true; }
            }
        }
        // mrbp_qmc/vmc.py:215-233 (recast to the supercell).  The forced
        // first yield (d = 0): the reference takes `ini_sys_conf` as it
        // comes; the pair sums here want positions inside [0, L), so they get
        // the image inside the box (and the one-body factor the position as
        // given, below).  The stored configuration stays as the caller gave
        // it until the first accepted move.
        zn[p] = wrap_box(zp + d, m.L);
        if (forced) outside = outside || zn[p] != zp;
    }
    QMC_SECTION("resort");
    // (one odd-even pass every QMC_RESORT_EVERY steps keeps the lanes sorted
    // well enough: a particle moves a few per cent of the spacing per step)
    // exactly ascending lanes: the fast pair sum of qmc_sorted64.h
    // (33 <= N <= 63 as well: the ring then has N members)
    constexpr bool S64 = QMC_SORTED64 && G == 64 && P == 1 && !ZC;
    // (66 <= N <= 126, N even, as well: N / 2 lanes with two particles each)
    constexpr bool S128 = QMC_SORTED128 && G == 64 && P == 2 && !ZC;
    bool fast = false;
    // (an initial configuration with a particle outside the box: the row in
    // memory is ordered by the positions as given, not by their images, and
    // the one-body factor wants the position as given -- the general path,
    // which takes any order, evaluates it; wave-uniform, first yield only)
    if (forced)
        outside = __builtin_amdgcn_ballot_w64(outside) != 0ull;
    if constexpr (S64) {
        fast = !m.is_ideal && !outside &&
               sort_lanes64(zn[0], labn[0], gl, PAD ? n : 64) &&
               (PAD ? far_partner_ok_ring(m, zn[0], gl, n)
                    : far_partner_ok64(m, zn[0], gl));
    } else if constexpr (S128) {
        fast = !m.is_ideal && !outside && (!PAD || (n & 1) == 0) &&
               sort_rows128(zn, labn, gl, PAD ? n / 2 : 64) &&
               (PAD ? far_partner_ok_ring128(m, zn, gl, n / 2)
                    : far_partner_ok128(m, zn, gl));
    }
    if constexpr (S64 || S128) {
        // (diagnostic, cold path only: walkers that leave the sorted-row path)
        if (!fast && !m.is_ideal && gl == 0 && active)
            atomicAdd(m.diag, 1ull);
    } else if constexpr (G == 64 && P == 1 && QMC_LINEAR_ORDER) {
        // ascending order, wrap point anchored at the lane seam (qmc_device.h)
        // (one particle per lane: lanes 0 .. n-1 hold them)
        if (!forced) {
            anchor_seam(zn[0], labn[0], n);
            if ((a.step % QMC_RESORT_EVERY) == 0)
                resort_linear<G>(zn[0], labn[0], gl,
                                 a.step / QMC_RESORT_EVERY, n);
        }
    } else if constexpr (SlotMap<G, P>::CONSECUTIVE && QMC_LINEAR_ORDER) {
        if (!forced) {
            anchor_seam_rows<P>(zn, labn, n);
            if ((a.step % QMC_RESORT_EVERY) == 0)
                resort_linear_rows<P>(zn, labn, gl, a.step / QMC_RESORT_EVERY, n);
        }
    } else if (!forced && (a.step % QMC_RESORT_EVERY) == 0)
        resort_step<G, P>(zn, labn, gl, a.step / QMC_RESORT_EVERY, n, m.L,
                          m.half_L, lanes_in_use<G, PAD>(m));
    constexpr bool TWO_PASS = QMC_VMC_TWO_PASS && (G == 64);
    double F[P], ei[P], e_new = 0.0, wf_new;
    if (S64 && fast) {
        if constexpr (S64)
            eval_sorted64<R, true, !TWO_PASS, false, PAD>(
                m, zn[0], gl, lds, F[0], e_new, wf_new);
    } else if (S128 && fast) {
        if constexpr (S128)
            eval_sorted128<R, true, !TWO_PASS, false, PAD>(m, zn, gl, lds, F,
                                                           e_new, wf_new);
    } else {
        // (the general path; z1: see eval_walker -- the stored positions on
        // a first yield given outside the box, nothing was reordered then)
        double z1[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            int i = lane_particle<G, P, PAD>(m, gl, p);
            z1[p] = (outside && i < n) ? a.pos[wr * n + i] : zn[p];
        }
        if constexpr (TWO_PASS)
            eval_walker<G, P, PAD, true, false, ZC, R, false, false>(
                m, zn, z1, gl, lds, F, ei, e_new, wf_new);
        else
            eval_walker<G, P, PAD, true, false, ZC, R>(m, zn, z1, gl, lds, F,
                                                       ei, e_new, wf_new);
    }
    QMC_SECTION("metropolis+store");
    if (!forced) {
        if (!LEAN && a.tape) {
            ua = a.tape[(wr * a.tape_steps + a.tape_idx) * (n + 1) + n];
        } else if (!LEAN && a.gaussian) {
            double u1;
            philox_uniform2(a.seed, slot, a.step, 0u, STREAM_VMC_ACCEPT, ua,
                            u1);
        } else {
            // exactly one lane of the group holds particle 0 and one particle 1:
            // find them with a ballot each and read their words
            const unsigned long long bal0 = __ballot(has0);
            const unsigned long long bal1 = __ballot(has1);
            unsigned int hi, lo;
            if (G == 64) {
                const int src0 = __builtin_amdgcn_readfirstlane(
                    (int)__ffsll((long long)bal0) - 1) & 63;
                const int src1 = __builtin_amdgcn_readfirstlane(
                    (int)__ffsll((long long)bal1) - 1) & 63;
                hi = (unsigned)__builtin_amdgcn_readlane((int)aw0, src0);
                lo = (unsigned)__builtin_amdgcn_readlane((int)aw1, src1);
            } else {
                const int base = (threadIdx.x & 63) - gl;
                const unsigned long long gmask = (1ull << (G & 63)) - 1ull;
                const int src0 = base + ((__ffsll((long long)((bal0 >> base) &
                                                              gmask)) - 1) & (G - 1));
                const int src1 = base + ((__ffsll((long long)((bal1 >> base) &
                                                              gmask)) - 1) & (G - 1));
                hi = (unsigned)__shfl((int)aw0, src0, 64);
                lo = (unsigned)__shfl((int)aw1, src1, 64);
            }
            ua = u53(hi, lo);
        }
    }
    if (!active) return;
    // Metropolis test (qmc_base/vmc.py:636)
    // log(u) <= 0: an uphill move needs no logarithm (wave-uniform when one
    // wavefront owns one chain)
    bool acc = forced || ua <= 0.0 || wf_new > wf_cur;
    if (!acc) acc = wf_new > 0.5 * log_pos(ua) + wf_cur;
    // (one chain per wavefront: every lane holds the same decision)
    if (G == 64) acc = __builtin_amdgcn_readfirstlane((int)acc) != 0;
    if (acc) {
        if constexpr (TWO_PASS) {
            QMC_SECTION("energy_pass");
            QMC_SECTION_PHASE(QMC_NSEC / 2);
            double wf_unused;
            if (S64 && fast) {
                if constexpr (S64)
                    eval_sorted64<R, false, true, true, PAD>(
                        m, zn[0], gl, lds, F[0], e_new, wf_unused);
            } else if (S128 && fast) {
                if constexpr (S128)
                    eval_sorted128<R, false, true, true, PAD>(
                        m, zn, gl, lds, F, e_new, wf_unused);
            } else {
                double z1[P];
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    int i = lane_particle<G, P, PAD>(m, gl, p);
                    z1[p] = (outside && i < n) ? a.pos[wr * n + i] : zn[p];
                }
                eval_walker<G, P, PAD, false, false, ZC, R, true, true>(
                    m, zn, z1, gl, lds, F, ei, e_new, wf_unused);
            }
            QMC_SECTION_PHASE(0);
            QMC_SECTION("store");
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            int i = lane_particle<G, P, PAD>(m, gl, p);
            if (i < n && !forced) {
                a.pos[w * n + i] = zn[p];
                a.label[w * n + i] = (unsigned short)labn[p];
            }
        }
        // (the initial yield too: log|psi| of the initial state then comes
        // from the same pair sum as every later one -- the sorted-row path on
        // its shapes -- instead of qmc_vmc_set_state's batch evaluation)
        wf_cur = wf_new;
        e_cur = e_new;       // energy only re-evaluated on accepted moves
    }                        // (qmc_base/jastrow/vmc.py:253-262)
    if (!LEAN && a.ser_pos) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            int i = lane_particle<G, P, PAD>(m, gl, p);
            // series in the original particle order (a rejected move leaves
            // pos / label as they were: read them back)
            if (i < n) {
                // (the forced first yield: the configuration as it was given)
                const bool moved = acc && !forced;
                const int lb = moved ? labn[p] : (int)a.label[w * n + i];
                a.ser_pos[(a.y * a.W + w) * n + lb] =
                    moved ? zn[p] : a.pos[w * n + i];
            }
        }
    }
    if (gl == 0) {
        if (G != 64 && !a.reset_sums) {
            se = a.sum_e[w]; se2 = a.sum_e2[w]; na = a.n_acc[w];
        }
        a.wf[w] = wf_cur;
        a.ecarry[w] = e_cur;
        a.sum_e[w] = se + e_cur;
        a.sum_e2[w] = fma(e_cur, e_cur, se2);
        a.n_acc[w] = na + (acc ? 1 : 0);
        if (!LEAN) {
            if (a.ser_wf) a.ser_wf[a.y * a.W + w] = wf_cur;
            if (a.ser_e) a.ser_e[a.y * a.W + w] = e_cur;
            if (a.ser_stat) a.ser_stat[a.y * a.W + w] = acc ? 1 : 0;
        }
    }
    QMC_SECTION("end");
}

// ---- DMC ---------------------------------------------------------------
// Device-resident control block of a DMC ensemble.
struct DmcCtl {
    long long prev_nw;      // walkers in the parent buffer
    long long nw;           // walkers after branching (this step)
    double ref_energy;
    double total_energy, total_weight;
    double e_t, w_t;        // estimators of this step (local or global)
    long long spare_nw;     // slots holding a valid spare normal
    unsigned int step;
    unsigned int pad;
};

struct EvolveArgs {
    const double *ppos, *pdrift, *penergy;   // parents
    double *cpos, *cdrift, *cenergy, *cweight; // children (cweight: log-weights)
    const unsigned short *plabel;             // parents' lane -> particle index
    unsigned short *clabel;
    double *eslot;            // energy the slot held in the previous iteration
    const long long *ref;
    const DmcCtl *ctl;
    const double *g_tape;     // [slot][N] standard normals or null
    double *spare;            // [maxw][N] second Box-Muller normal of a pair
    long long maxw;
    double dt, sigma;
    unsigned long long seed;
    unsigned int slot0;
    int fix_stale;
};

// Drift-diffusion + local energy of every child walker
// (qmc_base/jastrow/dmc.py:758-825, 892-942).
template <int G, int P, bool PAD, bool ZC, typename R = double>
__global__ void __launch_bounds__(WalkBlock<G>::N QMC_LB_WAVES_DMC)
dmc_evolve_kernel(const DevModel *__restrict__ mp, EvolveArgs a)
{
    // model constants live in device memory: scalar loads on demand keep the
    // SGPR file free for the hot loop (by-value they overflow it)
    const DevModel &m = *mp;
    [[maybe_unused]] constexpr int QMC_SEC_OFF = 0;
    extern __shared__ double smem[];
    constexpr int GPB = WalkBlock<G>::N / G;
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    static_assert(StepLds<G, P, PAD, ZC, true>::DOUBLES >=
                      GroupLds<G, P, ZC>::DOUBLES,
                  "eval_walker is the fallback on the same LDS region");
    double *lds = smem + (size_t)grp * StepLds<G, P, PAD, ZC, true>::DOUBLES;
    const long long s = walker_of_block<GPB>(blockIdx.x, grp);
    const long long nw = a.ctl->nw;
    // whole block beyond the population: nothing to do
    if ((GPB == 1 ? s : (long long)blockIdx.x * GPB) >= nw) return;
    QMC_SECTION("top");
    const bool active = s < nw;
    const long long sr = active ? s : 0;
    const int n = m.n;
    const unsigned int step = a.ctl->step;
    const double ref_energy = a.ctl->ref_energy;
    // slots that existed at the previous (even) step have a stored normal
    const long long spare_nw = a.ctl->spare_nw;
    // (one walker per wavefront: the slot index is wave-uniform, the parent id
    // and the two energies the weight needs after the pair sum are scalar loads
    // issued here, their latency under everything else)
    const long long su = (G == 64) ? qmc_uniform(sr) : sr;
    const long long par = a.ref[su];
    double e_par = 0.0, e_slot = 0.0;
    if (G == 64) {
        e_par = a.penergy[par];
        e_slot = a.eslot[su];
    }

    QMC_SECTION("load+philox+wrap");
    double z[P];
    int lab[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int i = lane_particle<G, P, PAD>(m, gl, p);
        double zz = 0.0;
        lab[p] = i;
        if (i < n) {
            double z0 = a.ppos[par * n + i];
            double f0 = a.pdrift[par * n + i];
            // random numbers belong to the particle (label), not to the lane
            const int li = (int)a.plabel[par * n + i];
            lab[p] = li;
            double g;
            if (a.g_tape) {
                g = a.g_tape[sr * n + li];
            } else if (DmcSpare<P>::ON && (step & 1u) && sr < spare_nw) {
                // odd step: the sine-branch normal stored by the even step
                g = a.spare[sr * n + li];
            } else {
                // time steps 2m, 2m+1 share one Philox block: cosine branch
                // now, sine branch kept for the next step of this slot
                double g0, g1;
                dmc_normal2(a.seed, a.slot0 + (unsigned)sr, step >> 1,
                            (unsigned)li, g0, g1);
                g = (step & 1u) ? g1 : g0;
                if (DmcSpare<P>::ON && !(step & 1u) && active)
                    a.spare[sr * n + li] = g1;
            }
            // ith_diffusion (qmc_base/jastrow/dmc.py:661-671)
            double zn = z0 + 2 * f0 * a.dt + a.sigma * g;
            zz = wrap_box(zn, m.L);
        }
        z[p] = zz;
    }
    QMC_SECTION("resort");
    constexpr bool S64 = QMC_SORTED64 && G == 64 && P == 1 && !ZC;
    constexpr bool S128 = QMC_SORTED128 && G == 64 && P == 2 && !ZC;
    bool fast = false;
    if constexpr (S64) {
        fast = !m.is_ideal && sort_lanes64(z[0], lab[0], gl, PAD ? n : 64) &&
               (PAD ? far_partner_ok_ring(m, z[0], gl, n)
                    : far_partner_ok64(m, z[0], gl));
    } else if constexpr (S128) {
        fast = !m.is_ideal && (!PAD || (n & 1) == 0) &&
               sort_rows128(z, lab, gl, PAD ? n / 2 : 64) &&
               (PAD ? far_partner_ok_ring128(m, z, gl, n / 2)
                    : far_partner_ok128(m, z, gl));
    }
    if constexpr (S64 || S128) {
        // (diagnostic, cold path only: walkers that leave the sorted-row path)
        if (!fast && !m.is_ideal && gl == 0 && active)
            atomicAdd(m.diag, 1ull);
    } else if constexpr (G == 64 && P == 1 && QMC_LINEAR_ORDER) {
        anchor_seam(z[0], lab[0], n);
        if ((step % QMC_RESORT_EVERY) == 0)
            resort_linear<G>(z[0], lab[0], gl, step / QMC_RESORT_EVERY, n);
    } else if constexpr (SlotMap<G, P>::CONSECUTIVE && QMC_LINEAR_ORDER) {
        anchor_seam_rows<P>(z, lab, n);
        if ((step % QMC_RESORT_EVERY) == 0)
            resort_linear_rows<P>(z, lab, gl, step / QMC_RESORT_EVERY, n);
    } else if ((step % QMC_RESORT_EVERY) == 0)
        resort_step<G, P>(z, lab, gl, step / QMC_RESORT_EVERY, n, m.L,
                          m.half_L, lanes_in_use<G, PAD>(m));
    // positions and labels leave now: not live across the pair sum
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int i = lane_particle<G, P, PAD>(m, gl, p);
        if (active && i < n) {
            a.cpos[s * n + i] = z[p];
            a.clabel[s * n + i] = (unsigned short)lab[p];
        }
    }
    double F[P], ei[P], e_next, wf;
    if (S64 && fast) {
        if constexpr (S64)
            eval_sorted64<R, false, true, false, PAD>(m, z[0], gl, lds, F[0],
                                                      e_next, wf);
    } else if (S128 && fast) {
        if constexpr (S128)
            eval_sorted128<R, false, true, false, PAD>(m, z, gl, lds, F,
                                                       e_next, wf);
    } else
        eval_walker<G, P, PAD, false, false, ZC, R>(m, z, z, gl, lds, F, ei,
                                                    e_next, wf);
    QMC_SECTION("weight+store");
    if (!active) return;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int i = lane_particle<G, P, PAD>(m, gl, p);
        if (i < n) a.cdrift[s * n + i] = F[p];
    }
    if (gl == 0) {
        if (G != 64) { e_par = a.penergy[par]; e_slot = a.eslot[s]; }
        // the reference averages with the energy slot s held in
        // the previous iteration (jastrow/dmc.py:810), not the parent's.
        double e_old = a.fix_stale ? e_par : e_slot;
        double mean_energy = (e_next + e_old) / 2;
        a.cenergy[s] = e_next;
        // The LOGARITHM of the branching weight: the exponential (qmc_base/
        // jastrow/dmc.py:818-825) is taken where the weight is used, by the one
        // thread per walker of the branching kernels.  Here it is one value per
        // wavefront, and a wavefront pays the instructions of exp() -- forty
        // of this kernel's 800 per walker-step at N = 64 -- whatever the
        // number of lanes that need the result.
        a.cweight[s] = -a.dt * (mean_energy - ref_energy);
        a.eslot[s] = e_par;
    }
    QMC_SECTION("end");
}
