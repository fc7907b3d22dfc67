// qmcwalk.hip -- kernels and C-ABI of libqmcwalk.so (gfx950 / MI355X).
// Interface: include/qmcwalk.h.  Device building blocks: qmc_device.h.
//
// HBM layout (all fp64, "slot-major"): a walker slot owns one contiguous row
// of N positions and one of N drifts, pos[W][N] / drift[W][N]; the lane that
// owns particle i of walker w reads pos[w][i], so a lane group loads its row
// with one coalesced 8*G-byte access.  Per-walker scalars (energy, weight,
// slot energy, parent index) are plain [W] arrays.  DMC keeps two such
// population buffers (parents / children) that swap roles every time step.
#include "qmc_device.h"
#include "../../include/qmcwalk.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

// --------------------------------------------------------------- errors ---
static thread_local std::string g_err;

static int fail(const std::string &msg)
{
    g_err = msg;
    return 1;
}

#define HIP_TRY(expr)                                                        \
    do {                                                                     \
        hipError_t _e = (expr);                                              \
        if (_e != hipSuccess)                                                \
            return fail(std::string(#expr) + ": " + hipGetErrorString(_e));  \
    } while (0)

extern "C" const char *qmc_last_error(void) { return g_err.c_str(); }
extern "C" int qmc_abi_version(void) { return QMCWALK_ABI_VERSION; }
extern "C" int qmc_device_count(int *count)
{
    HIP_TRY(hipGetDeviceCount(count));
    return 0;
}

// ------------------------------------------------------------- handles ----
struct qmc_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    DevModel dm;
    DevModel *dm_dev = nullptr;
    qmc_model_params mp;
    int G = 64, P = 1;
    bool pad = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

static constexpr int BLOCK = 256;          // 4 wavefronts per workgroup

static int pick_shape(int n, int &G, int &P, bool &pad)
{
    if (n < 1) return 1;
    if (n <= 16) { G = 16; P = 1; }
    else if (n <= 32) { G = 16; P = 2; }
    else if (n <= 64) { G = 64; P = 1; }
    else if (n <= 128) { G = 64; P = 2; }
    else if (n <= 256) { G = 64; P = 4; }
    else if (n <= 512) { G = 64; P = 8; }
    else return 1;
    pad = (n != G * P);
    return 0;
}

// --------------------------------------------------------------- kernels ---
struct EvalArgs {
    const double *pos;   // [W][N]
    double *wf, *energy; // [W]
    double *ith, *drift; // [W][N]
    long long nconf;
};

template <int G, int P, bool PAD, bool ZC>
__global__ void __launch_bounds__(BLOCK)
evaluate_kernel(const DevModel *__restrict__ mp, EvalArgs a)
{
    // model constants live in device memory: scalar loads on demand keep the
    // SGPR file free for the hot loop (by-value they overflow it)
    const DevModel &m = *mp;
    extern __shared__ double smem[];
    constexpr int GPB = BLOCK / G;            // groups per block
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    double *lds = smem + (size_t)grp * GroupLds<G, P, ZC>::DOUBLES;
    const long long w = (long long)blockIdx.x * GPB + grp;
    const bool active = w < a.nconf;
    const long long wr = active ? w : 0;
    double z[P], F[P], ei[P], E, wf;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int i = gl + G * p;
        z[p] = (i < m.n) ? a.pos[wr * m.n + i] : 0.0;
    }
    eval_walker<G, P, PAD, true, true, ZC>(m, z, gl, lds, F, ei, E, wf);
    if (!active) return;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int i = gl + G * p;
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

template <int G, int P, bool PAD, bool ZC>
__global__ void __launch_bounds__(BLOCK)
prepare_kernel(const DevModel *__restrict__ mp, PrepArgs a)
{
    // model constants live in device memory: scalar loads on demand keep the
    // SGPR file free for the hot loop (by-value they overflow it)
    const DevModel &m = *mp;
    extern __shared__ double smem[];
    constexpr int GPB = BLOCK / G;
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    double *lds = smem + (size_t)grp * GroupLds<G, P, ZC>::DOUBLES;
    const long long w = (long long)blockIdx.x * GPB + grp;
    const bool active = w < a.nconf;
    const long long wr = active ? w : 0;
    double z[P], F[P], ei[P], E, wf;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int i = gl + G * p;
        z[p] = (i < m.n) ? a.pos[wr * m.n + i] : 0.0;
    }
    eval_walker<G, P, PAD, false, false, ZC>(m, z, gl, lds, F, ei, E, wf);
    if (!active) return;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int i = gl + G * p;
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
template <int G, int P, bool PAD, bool ZC, bool LEAN>
__global__ void __launch_bounds__(BLOCK)
vmc_step_kernel(const DevModel *__restrict__ mp, VmcArgs a)
{
    const DevModel &m = *mp;
    extern __shared__ double smem[];
    constexpr int GPB = BLOCK / G;
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    double *lds = smem + (size_t)grp * GroupLds<G, P, ZC>::DOUBLES;
    const long long w = (long long)blockIdx.x * GPB + grp;
    const bool active = w < a.W;
    const long long wr = active ? w : 0;
    const int n = m.n;
    const unsigned int slot = a.chain0 + (unsigned int)wr;
    // The very first yield of a generator is the initial state itself,
    // flagged ACCEPTED (qmc_base/vmc.py:616-618): a forced zero move.
    const bool forced = a.forced != 0;

    double zn[P];
    int labn[P];              // original particle index held by each lane
    double ua = 1.0;          // accept uniform (particle 0's spare double)
    double mine = -1.0;       // >= 0 only in the lane that holds particle 0
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int i = gl + G * p;
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
                double u0, u1;
                philox_uniform2(a.seed, slot, a.step, li, STREAM_VMC_MOVE, u0,
                                u1);
                d = (u0 - 0.5) * a.move_spread;
                // the accept draw is the spare double of particle 0
                mine = (li == 0u) ? u1 : mine;
            }
        }
        // mrbp_qmc/vmc.py:215-233 (recast to the supercell)
        zn[p] = forced ? zp : wrap_box(zp + d, m.L);
    }
    if (!forced) resort_step<G, P>(zn, labn, gl, a.step, n, m.L, m.half_L);
    double F[P], ei[P], e_new, wf_new;
    eval_walker<G, P, PAD, true, false, ZC>(m, zn, gl, lds, F, ei, e_new,
                                            wf_new);
    if (!forced) {
        if (!LEAN && a.tape) {
            ua = a.tape[(wr * a.tape_steps + a.tape_idx) * (n + 1) + n];
        } else if (!LEAN && a.gaussian) {
            double u1;
            philox_uniform2(a.seed, slot, a.step, 0u, STREAM_VMC_ACCEPT, ua,
                            u1);
        } else {
            // exactly one lane of the group holds particle 0 (ua >= 0 there):
            // find it with a ballot and read its value
            const unsigned long long bal = __ballot(mine >= 0.0);
            if (G == 64) {
                const int src = __builtin_amdgcn_readfirstlane(
                    (int)__ffsll((long long)bal) - 1) & 63;
                int lo = __double2loint(mine), hi = __double2hiint(mine);
                lo = __builtin_amdgcn_readlane(lo, src);
                hi = __builtin_amdgcn_readlane(hi, src);
                ua = __hiloint2double(hi, lo);
            } else {
                const int base = (threadIdx.x & 63) - gl;
                const unsigned long long grp_bits =
                    (bal >> base) & ((1ull << (G & 63)) - 1ull);
                const int src = base + ((__ffsll((long long)grp_bits) - 1) & (G - 1));
                ua = __shfl(mine, src, 64);
            }
        }
    }
    if (!active) return;
    double wf_cur = a.wf[w];
    double e_cur = a.ecarry[w];
    // Metropolis test (qmc_base/vmc.py:636)
    // log(u) <= 0: an uphill move needs no logarithm (wave-uniform when one
    // wavefront owns one chain)
    bool acc = forced || ua <= 0.0 || wf_new > wf_cur;
    if (!acc) acc = wf_new > 0.5 * log_pos(ua) + wf_cur;
    if (acc) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            int i = gl + G * p;
            if (i < n && !forced) {
                a.pos[w * n + i] = zn[p];
                a.label[w * n + i] = (unsigned short)labn[p];
            }
        }
        if (!forced) wf_cur = wf_new;
        e_cur = e_new;       // energy only re-evaluated on accepted moves
    }                        // (qmc_base/jastrow/vmc.py:253-262)
    if (!LEAN && a.ser_pos) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            int i = gl + G * p;
            // series in the original particle order (a rejected move leaves
            // pos / label as they were: read them back)
            if (i < n) {
                const int lb = acc ? labn[p] : (int)a.label[w * n + i];
                a.ser_pos[(a.y * a.W + w) * n + lb] =
                    acc ? zn[p] : a.pos[w * n + i];
            }
        }
    }
    if (gl == 0) {
        double se = a.reset_sums ? 0.0 : a.sum_e[w];
        double se2 = a.reset_sums ? 0.0 : a.sum_e2[w];
        long long na = a.reset_sums ? 0 : a.n_acc[w];
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

struct BranchArgs {
    const double *weight;     // parent weights [maxw]
    const double *energy;     // parent energies [maxw]
    int *count;               // clone counts [maxw]
    long long *block_tot;     // [nblocks]
    long long *block_off;     // [nblocks]
    double *block_esum;       // [nblocks] partial sums of parent energies
    long long *ref;           // cloning table [maxw]
    DmcCtl *ctl;
    const double *u_tape;     // uniforms of this step or null
    long long maxw;
    unsigned long long seed;
    unsigned int slot0;
};

static constexpr int BR_ITEMS = 4;                    // parents per thread
static constexpr int BR_TILE = BLOCK * BR_ITEMS;      // parents per block

// Clone counts c_s = int(w_s + u_s) (qmc_base/dmc.py:641-642) + block totals.
// (`tile` = blockIdx.x in the multi-block kernels; the fused small-population
// kernel walks the tiles with one workgroup)
__device__ __forceinline__ void branch_count_tile(const BranchArgs &a, int tile)
{
    __shared__ long long red[BLOCK / 64];
    const long long prev_nw = a.ctl->prev_nw;
    const unsigned int step = a.ctl->step;
    long long base = (long long)tile * BR_TILE + threadIdx.x * BR_ITEMS;
    long long tot = 0;
#pragma unroll
    for (int k = 0; k < BR_ITEMS; ++k) {
        long long s = base + k;
        int c = 0;
        if (s < prev_nw) {
            double u, u1;
            if (a.u_tape) u = a.u_tape[s];
            else philox_uniform2(a.seed, a.slot0 + (unsigned)s, step, 0u,
                                 STREAM_DMC_BRANCH, u, u1);
            c = (int)(a.weight[s] + u);
            a.count[s] = c;
        }
        tot += c;
    }
    for (int msk = 1; msk < 64; msk <<= 1)
        tot += __shfl_xor(tot, msk, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int i = 0; i < BLOCK / 64; ++i) t += red[i];
        a.block_tot[tile] = t;
    }
    __syncthreads();
}

__global__ void __launch_bounds__(BLOCK) branch_count_kernel(BranchArgs a)
{
    branch_count_tile(a, (int)blockIdx.x);
}

// Scatter parent indices into the cloning table in parent order, truncated at
// max_num_walkers; per-block partial sums of the yielded energies
// E_t = sum_s E_parent(ref[s]) (qmc_base/dmc.py:759-762).
__device__ __forceinline__ void branch_scatter_tile(const BranchArgs &a, int tile,
                                                    long long tile_off)
{
    __shared__ long long wtot[BLOCK / 64];
    __shared__ double wsum[BLOCK / 64];
    const long long prev_nw = a.ctl->prev_nw;
    long long base = (long long)tile * BR_TILE + threadIdx.x * BR_ITEMS;
    int c[BR_ITEMS];
    long long mine = 0;
#pragma unroll
    for (int k = 0; k < BR_ITEMS; ++k) {
        long long s = base + k;
        c[k] = (s < prev_nw) ? a.count[s] : 0;
        mine += c[k];
    }
    // exclusive scan of `mine` over the block
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    long long incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
        long long t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    long long woff = 0;
    for (int i = 0; i < wv; ++i) woff += wtot[i];
    long long off = tile_off + woff + incl - mine;
    double esum = 0.0;
#pragma unroll
    for (int k = 0; k < BR_ITEMS; ++k) {
        long long s = base + k;
        long long lo = off, hi = off + c[k];
        if (hi > a.maxw) hi = a.maxw;
        for (long long t = lo; t < hi; ++t) a.ref[t] = s;
        if (hi > lo) esum += (double)(hi - lo) * a.energy[s];
        off += c[k];
    }
    for (int msk = 1; msk < 64; msk <<= 1)
        esum += __shfl_xor(esum, msk, 64);
    if (lane == 0) wsum[wv] = esum;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < BLOCK / 64; ++i) t += wsum[i];
        a.block_esum[tile] = t;
    }
    __syncthreads();
}

// One workgroup per tile of parents.  The tile's offset into the cloning table
// is the sum of the clone totals of the tiles before it (at most maxw / 1024
// values, summed here by the workgroup itself: no separate scan launch); the
// last tile in use also owns the capped population size
// (qmc_base/dmc.py:638-653).
__global__ void __launch_bounds__(BLOCK) branch_scatter_kernel(BranchArgs a)
{
    __shared__ long long part[BLOCK / 64];
    const int tile = (int)blockIdx.x;
    const long long prev_nw = a.ctl->prev_nw;
    const int used = (int)((prev_nw + BR_TILE - 1) / BR_TILE);
    if (tile >= used) {
        if (tile == 0 && threadIdx.x == 0) a.ctl->nw = 0;   // extinct
        return;
    }
    long long t = 0;
    for (int i = threadIdx.x; i < tile; i += BLOCK) t += a.block_tot[i];
    for (int msk = 1; msk < 64; msk <<= 1) t += __shfl_xor(t, msk, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = t;
    __syncthreads();
    long long tile_off = 0;
    for (int i = 0; i < BLOCK / 64; ++i) tile_off += part[i];
    if (tile == used - 1 && threadIdx.x == 0) {
        const long long total = tile_off + a.block_tot[tile];
        a.ctl->nw = total < a.maxw ? total : a.maxw;
    }
    __syncthreads();
    branch_scatter_tile(a, tile, tile_off);
}

// Small populations (at most BR_FUSED_TILES tiles of 1024 parents, i.e. the
// reference's default 480 / 512 walkers): the whole branching step -- counts,
// scan, cloning table, E_t and W_t -- in ONE workgroup.  There a time step
// costs the device-side latency of its dependent launches (about 3 us each),
// not their work: 6 launches -> 3, 19.8 -> 15.0 us per step at 480 walkers.
// (Beyond two tiles the serial walk over the tiles loses: 4096 walkers
// 22 -> 32 us.)
static constexpr int BR_FUSED_TILES = 2;

__global__ void __launch_bounds__(BLOCK)
branch_fused_kernel(BranchArgs a, double *partial)
{
    const long long prev_nw = a.ctl->prev_nw;
    const int used = (int)((prev_nw + BR_TILE - 1) / BR_TILE);
    for (int tile = 0; tile < used; ++tile) branch_count_tile(a, tile);
    if (threadIdx.x == 0) {
        long long run = 0;
        for (int i = 0; i < used; ++i) {
            const long long v = a.block_tot[i];
            a.block_off[i] = run;
            run += v;
        }
        a.ctl->nw = run < a.maxw ? run : a.maxw;
    }
    __syncthreads();
    for (int tile = 0; tile < used; ++tile)
        branch_scatter_tile(a, tile, a.block_off[tile]);
    if (threadIdx.x == 0) {
        double e_t = 0.0;
        for (int i = 0; i < used; ++i) e_t += a.block_esum[i];
        const double w_t = (double)a.ctl->nw;     // unit weights after branching
        a.ctl->e_t = e_t;
        a.ctl->w_t = w_t;
        if (partial) { partial[0] = e_t; partial[1] = w_t; }
    }
}

// Sum of the per-tile energy partials in a fixed order (one workgroup of
// BLOCK threads; the same order wherever it is used) -> sh[0].
__device__ __forceinline__ double sum_block_esum(const double *block_esum,
                                                 const DmcCtl *ctl, double *sh)
{
    const long long prev_nw = ctl->prev_nw;
    const int used = (int)((prev_nw + BR_TILE - 1) / BR_TILE);
    double t = 0.0;
    for (int i = threadIdx.x; i < used; i += BLOCK) t += block_esum[i];
    sh[threadIdx.x] = t;
    __syncthreads();
    for (int off = BLOCK / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    return sh[0];
}

// This rank's E_t, W_t for the external (multi-GPU) reduction.
__global__ void __launch_bounds__(BLOCK)
dmc_local_sums_kernel(const double *block_esum, DmcCtl *ctl, double *partial)
{
    __shared__ double sh[BLOCK];
    sum_block_esum(block_esum, ctl, sh);
    if (threadIdx.x == 0) {
        ctl->e_t = sh[0];
        ctl->w_t = (double)ctl->nw;     // unit weights after branching
        if (partial) { partial[0] = sh[0]; partial[1] = (double)ctl->nw; }
    }
}

struct EvolveArgs {
    const double *ppos, *pdrift, *penergy;   // parents
    double *cpos, *cdrift, *cenergy, *cweight; // children
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
template <int G, int P, bool PAD, bool ZC>
__global__ void __launch_bounds__(BLOCK)
dmc_evolve_kernel(const DevModel *__restrict__ mp, EvolveArgs a)
{
    // model constants live in device memory: scalar loads on demand keep the
    // SGPR file free for the hot loop (by-value they overflow it)
    const DevModel &m = *mp;
    extern __shared__ double smem[];
    constexpr int GPB = BLOCK / G;
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    double *lds = smem + (size_t)grp * GroupLds<G, P, ZC>::DOUBLES;
    const long long s = (long long)blockIdx.x * GPB + grp;
    const long long nw = a.ctl->nw;
    // whole block beyond the population: nothing to do
    if ((long long)blockIdx.x * GPB >= nw) return;
    const bool active = s < nw;
    const long long sr = active ? s : 0;
    const int n = m.n;
    const unsigned int step = a.ctl->step;
    const double ref_energy = a.ctl->ref_energy;
    // slots that existed at the previous (even) step have a stored normal
    const long long spare_nw = a.ctl->spare_nw;
    const long long par = a.ref[sr];

    double z[P];
    int lab[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int i = gl + G * p;
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
            } else if ((step & 1u) && sr < spare_nw) {
                // odd step: the sine-branch normal stored by the even step
                g = a.spare[sr * n + li];
            } else {
                // time steps 2m, 2m+1 share one Philox block: cosine branch
                // now, sine branch kept for the next step of this slot
                double g0, g1;
                philox_normal2(a.seed, a.slot0 + (unsigned)sr, step >> 1,
                               (unsigned)li, STREAM_DMC_DIFFUSE, g0, g1);
                g = (step & 1u) ? g1 : g0;
                if (!(step & 1u) && active) a.spare[sr * n + li] = g1;
            }
            // ith_diffusion (qmc_base/jastrow/dmc.py:661-671)
            double zn = z0 + 2 * f0 * a.dt + a.sigma * g;
            zz = wrap_box(zn, m.L);
        }
        z[p] = zz;
    }
    resort_step<G, P>(z, lab, gl, step, n, m.L, m.half_L);
    // positions and labels leave now: not live across the pair sum
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int i = gl + G * p;
        if (active && i < n) {
            a.cpos[s * n + i] = z[p];
            a.clabel[s * n + i] = (unsigned short)lab[p];
        }
    }
    double F[P], ei[P], e_next, wf;
    eval_walker<G, P, PAD, false, false, ZC>(m, z, gl, lds, F, ei, e_next, wf);
    if (!active) return;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int i = gl + G * p;
        if (i < n) a.cdrift[s * n + i] = F[p];
    }
    if (gl == 0) {
        double e_par = a.penergy[par];
        // the reference averages with the energy slot s held in
        // the previous iteration (jastrow/dmc.py:810), not the parent's.
        double e_old = a.fix_stale ? e_par : a.eslot[s];
        double mean_energy = (e_next + e_old) / 2;
        a.cenergy[s] = e_next;
        a.cweight[s] = exp(-a.dt * (mean_energy - ref_energy));
        a.eslot[s] = e_par;
    }
}

struct FinishArgs {
    DmcCtl *ctl;
    const double *total;      // global (E_t, W_t) or null -> local values
    const double *block_esum; // per-tile partials to sum here (E_t, unit
                              // weights) or null -> ctl->e_t / w_t are set
    double *ser_e, *ser_w, *ser_ref, *ser_acc;
    unsigned long long *ser_nw;
    long long ser_idx;
    double kappa, dt, target;
};

// E_ref feedback (qmc_base/dmc.py:759-785) + per-step series.
__global__ void __launch_bounds__(BLOCK) dmc_finish_kernel(FinishArgs a)
{
    __shared__ double sh[BLOCK];
    DmcCtl *c = a.ctl;
    double e_sum = 0.0;
    if (a.block_esum) e_sum = sum_block_esum(a.block_esum, c, sh);
    if (threadIdx.x != 0) return;
    double e_t, w_t;
    if (a.total) { e_t = a.total[0]; w_t = a.total[1]; }
    else if (a.block_esum) { e_t = e_sum; w_t = (double)c->nw; }
    else { e_t = c->e_t; w_t = c->w_t; }
    c->total_energy += e_t;
    c->total_weight += w_t;
    double accum = c->total_energy / c->total_weight;
    double ref = accum - a.kappa * log(w_t / a.target) / a.dt;
    c->ref_energy = ref;
    c->e_t = e_t;
    c->w_t = w_t;
    if (a.ser_e) {
        a.ser_e[a.ser_idx] = e_t;
        a.ser_w[a.ser_idx] = w_t;
        a.ser_nw[a.ser_idx] = (unsigned long long)c->nw;
        a.ser_ref[a.ser_idx] = ref;
        a.ser_acc[a.ser_idx] = accum;
    }
    // an even step stored spare normals for slots [0, nw); they are consumed
    // by the next (odd) step and invalid afterwards
    c->spare_nw = (c->step & 1u) ? 0 : c->nw;
    c->prev_nw = c->nw;
    c->step += 1;
}

// ---- DMC estimators ------------------------------
// Evaluated on the yielded population of a step: walker s carries the
// configuration of its parent, parents[ref[s]] (qmc_base/dmc.py:773-780).
struct EstArgs {
    const double *ppos;       // parent positions [maxw][N]
    const long long *ref;     // cloning table
    const DmcCtl *ctl;
    const double *aux_prev;   // [maxw][K][C] per-walker parts one step ago
    double *aux_act;          // [maxw][K][C] per-walker parts of this step
    double *partial;          // [nblocks][K][C] block partial sums
    long long maxw;
    long long step_idx;       // index of the step inside the block
    long long pfw;            // forward-walking length
    int n;                    // particles
    int K;                    // modes or bins
    int pure;
    double scale;             // S(k): 4 / L (angle k_m z = (pi/2) * m * scale * z)
                              // density: bin size L / num_bins
};

static constexpr int EST_BLOCKS = 1024;
static constexpr int EST_MAXK = 256;        // modes / bins supported per call
static constexpr int EST_CH = EST_MAXK / 64;

// Static structure factor parts of every yielded walker:
// rho_m = sum_i exp(i k_m z_i), k_m = 2 pi m / L, parts (|rho_m|^2, Re, Im);
// mixed estimator or forward-walking transport through the cloning table
// (qmc_base/jastrow/dmc.py:363-461, 483-566).
//
// The sum over particles IS a contraction, so it runs on the matrix cores:
// with m = KD a + b,  exp(i m t_i) = F_a(i) E_b(i),  F_a = exp(i KD a t_i),
// E_b = exp(i b t_i), and  rho[a][b] = sum_i F_a(i) E_b(i)  is a
// (2 KD x N) x (N x 2 KD) real product over the particle index, accumulated
// with v_mfma_f64_16x16x4_f64 (K = 4 particles per instruction).  The two
// factor tables cost one sincos and KD - 1 complex rotations per particle
// instead of one sincos per (mode, particle): 2560 -> ~200 VALU instructions
// per walker at N = 64, 64 modes, plus 16 MFMAs on the otherwise idle matrix
// pipe.  KD = 8 packs Re/Im of both factors into one 16x16 tile (<= 64
// modes); KD = 16 uses four tiles (<= 256 modes).
typedef double v4d __attribute__((ext_vector_type(4)));

template <int KD>
struct SsfShape {
    static constexpr int CH = 32;                    // particles per chunk
    static constexpr int ROWS = 2 * KD;              // Re and Im rows
    static constexpr int RS = CH + 4;                // padded row stride
    static constexpr int WAVE_DOUBLES = 2 * ROWS * RS;
    static constexpr int NM = (KD == 8) ? 1 : 4;     // modes per lane
};

// Table of exp(i b t), b = 0..KD-1, of one particle: rows [0,KD) real parts,
// rows [KD,2KD) imaginary parts, column = particle slot.
template <int KD>
__device__ __forceinline__ void ssf_fill_table(double *T, int col, double u,
                                               bool valid)
{
    constexpr int RS = SsfShape<KD>::RS;
    double s1, c1;
    sincos_halfpi(u, s1, c1);
    double er = valid ? 1.0 : 0.0, ei = 0.0;
    T[col] = er;
    T[KD * RS + col] = 0.0;
#pragma unroll
    for (int b = 1; b < KD; ++b) {
        double nr = er * c1 - ei * s1;
        double ni = er * s1 + ei * c1;
        er = nr; ei = ni;
        T[b * RS + col] = er;
        T[(KD + b) * RS + col] = ei;
    }
}

template <int KD>
__global__ void __launch_bounds__(BLOCK) dmc_ssf_mfma_kernel(EstArgs a)
{
    using S = SsfShape<KD>;
    constexpr int CH = S::CH, RS = S::RS, NM = S::NM;
    extern __shared__ double smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *X = smem + (size_t)wave * S::WAVE_DOUBLES;   // F_a (rows of D)
    double *Y = X + S::ROWS * RS;                        // E_b (columns of D)
    // (VMC ensembles use the kernel without a cloning table: ref = identity,
    // population = maxw chains)
    const long long nw = a.ctl ? a.ctl->nw : a.maxw;
    const long long wstride = (long long)gridDim.x * (BLOCK / 64);
    const int quad = lane >> 4, idx = lane & 15;
    // modes owned by this lane when the results are handed out
    int mo[NM];
#pragma unroll
    for (int r = 0; r < NM; ++r)
        mo[r] = (KD == 8) ? lane : 16 * (quad + 4 * r) + idx;
    double acc[NM][3];
#pragma unroll
    for (int r = 0; r < NM; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 0.0;
    const bool accumulate = !a.pure || a.step_idx < a.pfw;
    for (long long s = (long long)blockIdx.x * (BLOCK / 64) + wave; s < nw;
         s += wstride) {
        const long long par = a.ref ? a.ref[s] : s;
        double re[NM], im[NM];
#pragma unroll
        for (int r = 0; r < NM; ++r) re[r] = im[r] = 0.0;
        if (accumulate) {
            v4d Drr = {0, 0, 0, 0}, Dri = {0, 0, 0, 0}, Dir = {0, 0, 0, 0},
                Dii = {0, 0, 0, 0};
            for (int c0 = 0; c0 < a.n; c0 += CH) {
                {
                    // lanes 0..31 build E of particle `lane`, lanes 32..63
                    // build F of particle `lane - 32`
                    const int pl = lane & 31;
                    const int i = c0 + pl;
                    const bool valid = i < a.n;
                    const double u = a.scale * (valid ? a.ppos[par * a.n + i] : 0.0);
                    ssf_fill_table<KD>(lane < 32 ? Y : X, pl,
                                       lane < 32 ? u : (double)KD * u, valid);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int left = a.n - c0;
                const int ngroups = (left >= CH ? CH : left + 3) / 4;
                for (int g = 0; g < ngroups; ++g) {
                    const int col = 4 * g + quad;       // particle of this k
                    if (KD == 8) {
                        const double xa = X[idx * RS + col];
                        const double yb = Y[idx * RS + col];
                        Drr = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, yb, Drr,
                                                                   0, 0, 0);
                    } else {
                        const double xr = X[idx * RS + col];
                        const double xi = X[(KD + idx) * RS + col];
                        const double yr = Y[idx * RS + col];
                        const double yi = Y[(KD + idx) * RS + col];
                        Drr = __builtin_amdgcn_mfma_f64_16x16x4f64(xr, yr, Drr, 0, 0, 0);
                        Dri = __builtin_amdgcn_mfma_f64_16x16x4f64(xr, yi, Dri, 0, 0, 0);
                        Dir = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, yr, Dir, 0, 0, 0);
                        Dii = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, yi, Dii, 0, 0, 0);
                    }
                }
                __builtin_amdgcn_wave_barrier();    // tables are rewritten next
            }
            if (KD == 8) {
                // one tile holds the four quadrants RR | RI / IR | II; element
                // (row, col) sits in lane (col, row & 3), register row >> 2
                // (f64 MFMA C/D map: col = lane & 15, row = (lane >> 4) + 4 reg)
                double *Dl = X;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Dl[(quad + 4 * r) * 16 + idx] = Drr[r];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int fa = lane >> 3, eb = lane & 7;
                re[0] = Dl[fa * 16 + eb] - Dl[(8 + fa) * 16 + 8 + eb];
                im[0] = Dl[fa * 16 + 8 + eb] + Dl[(8 + fa) * 16 + eb];
                __builtin_amdgcn_wave_barrier();
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    re[r] = Drr[r] - Dii[r];
                    im[r] = Dri[r] + Dir[r];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < NM; ++r) {
            double v0 = 0.0, v1 = 0.0, v2 = 0.0;
            if (mo[r] < a.K) {
                if (accumulate) {
                    v0 = fma(re[r], re[r], im[r] * im[r]);
                    v1 = re[r]; v2 = im[r];
                }
                if (a.pure) {
                    const double *pp = a.aux_prev + ((size_t)par * a.K + mo[r]) * 3;
                    v0 += pp[0]; v1 += pp[1]; v2 += pp[2];
                    double *ap = a.aux_act + ((size_t)s * a.K + mo[r]) * 3;
                    ap[0] = v0; ap[1] = v1; ap[2] = v2;
                }
            }
            acc[r][0] += v0; acc[r][1] += v1; acc[r][2] += v2;
        }
    }
    // fixed-order block reduction: waves 0..3 (the tables' LDS is reused),
    // then the reduce kernel sums the blocks in index order
    __syncthreads();
    double *red = smem;                    // [BLOCK/64][K][3]
#pragma unroll
    for (int r = 0; r < NM; ++r)
        if (mo[r] < a.K) {
            double *q = red + ((size_t)wave * a.K + mo[r]) * 3;
            q[0] = acc[r][0]; q[1] = acc[r][1]; q[2] = acc[r][2];
        }
    __syncthreads();
    for (int i = threadIdx.x; i < a.K * 3; i += BLOCK) {
        double t = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) t += red[(size_t)w * a.K * 3 + i];
        a.partial[(size_t)blockIdx.x * a.K * 3 + i] = t;
    }
}

// Density histogram of every slot, lane = bin.  Reproduces the reference:
// the mixed estimator keeps adding into the slot's alternating buffer, the
// pure one copies the previous buffer slot by slot (no cloning table) and
// adds the current histogram while step < pfw (mrbp_qmc/dmc.py:472-547,
// qmc_base/jastrow/dmc.py:238-302).
__global__ void __launch_bounds__(BLOCK) dmc_density_kernel(EstArgs a)
{
    __shared__ int hist[BLOCK / 64][EST_MAXK];
    __shared__ double red[BLOCK / 64][EST_MAXK];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long nw = a.ctl->nw;
    const long long wstride = (long long)gridDim.x * (BLOCK / 64);
    double acc[EST_CH];
#pragma unroll
    for (int c = 0; c < EST_CH; ++c) acc[c] = 0.0;
    const bool count_now = !a.pure || a.step_idx < a.pfw;
    for (long long s = (long long)blockIdx.x * (BLOCK / 64) + wave; s < a.maxw;
         s += wstride) {
        const bool live = s < nw;
        if (!live && !a.pure) break;      // mixed: dead slots keep their data
        for (int b = lane; b < a.K; b += 64) hist[wave][b] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (live && count_now) {
            const long long par = a.ref[s];
            for (int i = lane; i < a.n; i += 64) {
                int b = (int)floor(a.ppos[par * a.n + i] / a.scale);
                b = b < 0 ? 0 : (b >= a.K ? a.K - 1 : b);
                atomicAdd(&hist[wave][b], 1);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < EST_CH; ++c) {
            const int b = c * 64 + lane;
            if (c * 64 >= a.K) break;
            if (b < a.K) {
                const size_t o = (size_t)s * a.K + b;
                double v = (a.pure ? a.aux_prev[o] : a.aux_act[o]) +
                           (double)hist[wave][b];
                a.aux_act[o] = v;
                if (live) acc[c] += v;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int c = 0; c < EST_CH; ++c) red[wave][c * 64 + lane] = acc[c];
    __syncthreads();
    for (int b = threadIdx.x; b < a.K; b += BLOCK) {
        double t = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) t += red[w][b];
        a.partial[(size_t)blockIdx.x * a.K + b] = t;
    }
}

// iter[step][k][c] = (sum over blocks) / divisor.  Fixed summation order:
// eight contiguous segments of blocks summed in index order by eight threads
// (independent loads in flight), the segment sums then added in order.
__global__ void __launch_bounds__(256)
est_reduce_kernel(const double *__restrict__ partial, int nblocks, int KC,
                  double divisor, double *__restrict__ out)
{
    __shared__ double seg_sum[8][32];
    const int j = threadIdx.x & 31, seg = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + j;
    const int per = (nblocks + 7) / 8;
    const int b0 = seg * per, b1 = min(nblocks, b0 + per);
    double t = 0.0;
    if (idx < KC) {
        int b = b0;
        for (; b + 4 <= b1; b += 4) {
            double v0 = partial[(size_t)b * KC + idx];
            double v1 = partial[(size_t)(b + 1) * KC + idx];
            double v2 = partial[(size_t)(b + 2) * KC + idx];
            double v3 = partial[(size_t)(b + 3) * KC + idx];
            t += v0; t += v1; t += v2; t += v3;
        }
        for (; b < b1; ++b) t += partial[(size_t)b * KC + idx];
    }
    seg_sum[seg][j] = t;
    __syncthreads();
    if (seg == 0 && idx < KC) {
        double r = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) r += seg_sum[q][j];
        out[idx] = r / divisor;
    }
}


// Gather the yielded ("actual") configurations: confs[s] = parents[ref[s]].
__global__ void dmc_gather_state_kernel(const double *ppos,
                                        const double *pdrift,
                                        const unsigned short *plabel,
                                        const long long *ref, long long nw,
                                        int n, double *confs)
{
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nw * n) return;
    long long s = idx / n;
    int i = (int)(idx % n);
    long long p = ref[s];
    int li = plabel[p * n + i];          // back to the original particle order
    confs[(s * 2 + 0) * n + li] = ppos[p * n + i];
    confs[(s * 2 + 1) * n + li] = pdrift[p * n + i];
}

// Walker record of the population rebalance: pos[N], drift[N], label[N] (as
// doubles), energy, weight.
__global__ void pack_walkers_kernel(const double *pos, const double *drift,
                                    const unsigned short *label,
                                    const double *energy, const double *weight,
                                    long long first, long long count, int n,
                                    double *buf)
{
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int rec = 3 * n + 2;
    if (idx >= count * rec) return;
    long long s = idx / rec;
    int j = (int)(idx % rec);
    long long src = first + s;
    double v;
    if (j < n) v = pos[src * n + j];
    else if (j < 2 * n) v = drift[src * n + (j - n)];
    else if (j < 3 * n) v = (double)label[src * n + (j - 2 * n)];
    else if (j == 3 * n) v = energy[src];
    else v = weight[src];
    buf[idx] = v;
}

__global__ void unpack_walkers_kernel(double *pos, double *drift,
                                      unsigned short *label,
                                      double *energy, double *weight,
                                      long long first, long long count, int n,
                                      const double *buf)
{
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int rec = 3 * n + 2;
    if (idx >= count * rec) return;
    long long s = idx / rec;
    int j = (int)(idx % rec);
    long long dst = first + s;
    double v = buf[idx];
    if (j < n) pos[dst * n + j] = v;
    else if (j < 2 * n) drift[dst * n + (j - n)] = v;
    else if (j < 3 * n) label[dst * n + (j - 2 * n)] = (unsigned short)v;
    else if (j == 3 * n) energy[dst] = v;
    else weight[dst] = v;
}

// ------------------------------------------------------------ dispatch ----
template <template <int, int, bool, bool> class L, typename... A>
static int dispatch_shape(const qmc_engine *e, A &&...args)
{
    const bool zc = e->dm.zclass != 0;
    // The masked variant is also the leaner one in registers (its per-pair
    // guards stop the compiler from keeping several pairs in flight: 110-160
    // VGPRs against 134-282 at P = 4, 8), and occupancy is what the large
    // shapes lack; each kernel family says from which P it wants it.
    const bool masked = e->pad || L<16, 1, false, false>::want_mask(e->P);
#define QMC_CASE(g, p)                                                        \
    if (e->G == g && e->P == p) {                                             \
        if (masked) {                                                         \
            if (zc) return L<g, p, true, true>::run(e, args...);              \
            return L<g, p, true, false>::run(e, args...);                     \
        }                                                                     \
        if (zc) return L<g, p, false, true>::run(e, args...);                 \
        return L<g, p, false, false>::run(e, args...);                        \
    }
    QMC_CASE(16, 1) QMC_CASE(16, 2) QMC_CASE(64, 1) QMC_CASE(64, 2)
    QMC_CASE(64, 4) QMC_CASE(64, 8)
#undef QMC_CASE
    return fail("unsupported boson_number");
}

template <int G, int P, bool ZC>
static size_t lds_bytes()
{
    return (size_t)(BLOCK / G) * GroupLds<G, P, ZC>::DOUBLES * sizeof(double);
}

// Dynamic LDS above the default limit must be opted into per kernel.
template <typename K>
static void allow_lds(K kernel, size_t bytes)
{
    if (bytes > 48 * 1024)
        (void)hipFuncSetAttribute((const void *)kernel,
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)bytes);
}

template <int G>
static unsigned grid_for(long long nwalkers)
{
    const long long gpb = BLOCK / G;
    return (unsigned)((nwalkers + gpb - 1) / gpb);
}

template <int G, int P, bool PAD, bool ZC>
struct LaunchEval {
    static bool want_mask(int np) { return np >= 4; }
    static int run(const qmc_engine *e, const EvalArgs &a)
    {
        if (a.nconf <= 0) return 0;
        const size_t lds = lds_bytes<G, P, ZC>();
        allow_lds(evaluate_kernel<G, P, PAD, ZC>, lds);
        hipLaunchKernelGGL((evaluate_kernel<G, P, PAD, ZC>),
                           dim3(grid_for<G>(a.nconf)), dim3(BLOCK),
                           lds, e->stream, e->dm_dev, a);
        HIP_TRY(hipGetLastError());
        return 0;
    }
};

template <int G, int P, bool PAD, bool ZC>
struct LaunchPrep {
    static bool want_mask(int np) { return np >= 4; }
    static int run(const qmc_engine *e, const PrepArgs &a)
    {
        if (a.nconf <= 0) return 0;
        const size_t lds = lds_bytes<G, P, ZC>();
        allow_lds(prepare_kernel<G, P, PAD, ZC>, lds);
        hipLaunchKernelGGL((prepare_kernel<G, P, PAD, ZC>),
                           dim3(grid_for<G>(a.nconf)), dim3(BLOCK),
                           lds, e->stream, e->dm_dev, a);
        HIP_TRY(hipGetLastError());
        return 0;
    }
};

template <int G, int P, bool PAD, bool ZC>
struct LaunchVmc {
    static bool want_mask(int np) { return np >= 4; }
    static int run(const qmc_engine *e, const VmcArgs &a)
    {
        const size_t lds = lds_bytes<G, P, ZC>();
        const bool lean = !a.tape && !a.gaussian && !a.ser_wf && !a.ser_e &&
                          !a.ser_stat && !a.ser_pos;
        if (lean) {
            allow_lds(vmc_step_kernel<G, P, PAD, ZC, true>, lds);
            hipLaunchKernelGGL((vmc_step_kernel<G, P, PAD, ZC, true>),
                               dim3(grid_for<G>(a.W)), dim3(BLOCK), lds,
                               e->stream, e->dm_dev, a);
        } else {
            allow_lds(vmc_step_kernel<G, P, PAD, ZC, false>, lds);
            hipLaunchKernelGGL((vmc_step_kernel<G, P, PAD, ZC, false>),
                               dim3(grid_for<G>(a.W)), dim3(BLOCK), lds,
                               e->stream, e->dm_dev, a);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
};

template <int G, int P, bool PAD, bool ZC>
struct LaunchEvolve {
    // P = 8 is capped at two waves per SIMD by LDS either way: unmasked
    // (186 VGPRs) it saves the guards
    static bool want_mask(int np) { return np == 4; }
    static int run(const qmc_engine *e, const EvolveArgs &a)
    {
        const size_t lds = lds_bytes<G, P, ZC>();
        allow_lds(dmc_evolve_kernel<G, P, PAD, ZC>, lds);
        hipLaunchKernelGGL((dmc_evolve_kernel<G, P, PAD, ZC>),
                           dim3(grid_for<G>(a.maxw)), dim3(BLOCK),
                           lds, e->stream, e->dm_dev, a);
        HIP_TRY(hipGetLastError());
        return 0;
    }
};

// -------------------------------------------------------------- engine ----
static void build_dev_model(const qmc_model_params &p, DevModel &d)
{
    memset(&d, 0, sizeof(d));
    d.n = (int)p.boson_number;
    d.is_free = p.is_free != 0;
    d.is_ideal = p.is_ideal != 0;
    d.defects_sep = (int)(p.defects_sep > 0 ? p.defects_sep : 1);
    d.L = p.supercell_size;
    d.half_L = 0.5 * d.L;
    d.rm = fabs(p.tbf_contact_cutoff);
    d.L_minus_rm = d.L - d.rm;
    d.two_over_L = 2.0 / d.L;
    d.k2_2pi = p.param_k2 * (2.0 / QMC_PI);
    d.k2 = p.param_k2;
    d.k2sq = d.k2 * d.k2;
    double phi = p.param_k2 * p.param_r_off;
    d.cphi = cos(phi);
    d.sphi = sin(phi);
    d.m_k2cphi = -d.k2 * d.cphi;
    d.k2sphi = d.k2 * d.sphi;
    double th = p.param_k2 * d.L;
    d.cth = cos(th);
    d.sth = fabs(sin(th));
    d.sth_sign = sin(th) < 0.0 ? (int)0x80000000u : 0;
    {
        // angles added to k2 z_own for the four short-range cases
        const double ang[4] = { -phi, phi, phi - th, th - phi };
        for (int v = 0; v < 4; ++v) {
            d.var_cos[v] = cos(ang[v]);
            d.var_sin[v] = sin(ang[v]);
        }
        d.m_k2 = -d.k2;
    }
    d.sin_rm = (d.rm >= d.half_L) ? 1.0 : sin(QMC_PI * d.rm / d.L);
    // sin(pi r / L) is flat near r = L/2: classify from positions there
    d.zclass = d.rm > 0.45 * d.L;
    double pi_L = QMC_PI / d.L;
    d.a_long = pi_L * p.param_beta;
    d.b_long = pi_L * pi_L * p.param_beta;
    d.beta = p.param_beta;
    d.inv_beta = p.param_beta != 0.0 ? 1.0 / p.param_beta : 0.0;
    d.log_am = log(fabs(p.param_am));
    d.z_a = 1.0 / (1.0 + p.lattice_ratio);
    d.z_b = p.lattice_ratio / (1.0 + p.lattice_ratio);
    d.k1 = p.param_k1;
    d.k1_2pi = p.param_k1 * (2.0 / QMC_PI);
    d.kp1 = p.param_kp1;
    d.e0 = p.param_e0;
    d.v0 = p.lattice_depth;
    d.v0d = p.defect_magnitude;
    d.uniform_barrier = (d.defects_sep == 1 || d.v0d == d.v0) ? 1 : 0;
    d.v_barrier = (d.defects_sep == 1) ? d.v0d : d.v0;
    d.k1_half = 0.5 * p.param_k1;
    d.v0_minus_e0 = p.lattice_depth - p.param_e0;
    if (!d.is_free) {
        double sh = sinh(0.5 * sqrt(d.v0 - d.e0) * d.z_b);
        d.cf = sqrt(1 + d.v0 / d.e0 * sh * sh);
    } else {
        d.cf = 1.0;
    }
}

extern "C" int qmc_engine_create(const qmc_model_params *model, int device,
                                 void *stream, qmc_engine **out)
{
    if (!model || !out) return fail("qmc_engine_create: null argument");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail("qmc_engine_create: no HIP device");
    if (device < 0 || device >= ndev)
        return fail("qmc_engine_create: bad device index");
    qmc_engine *e = new qmc_engine();
    e->device = device;
    e->mp = *model;
    if (pick_shape((int)model->boson_number, e->G, e->P, e->pad)) {
        delete e;
        return fail("qmc_engine_create: boson_number must be in [1, 512]");
    }
    build_dev_model(*model, e->dm);
    // domain of the short-range kernel (qmc_device.h pair_core): the matching
    // conditions of mrbp_qmc/model.py:340-392 give phi = k2 r_off in (0, pi/2)
    // and k2 rm in (0, pi/2) for every repulsive model
    if (!e->dm.is_ideal &&
        (e->dm.sphi < 0.0 || e->dm.cphi < 0.0 ||
         e->dm.k2 * e->dm.rm >= 0.5 * QMC_PI || e->dm.k2 <= 0.0)) {
        delete e;
        return fail("qmc_engine_create: two-body parameters outside the "
                    "model's domain (need 0 < k2 rm < pi/2, 0 <= k2 r_off <= pi/2)");
    }
    HIP_TRY(hipSetDevice(device));
    if (stream) {
        e->stream = (hipStream_t)stream;
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        e->own_stream = true;
    }
    HIP_TRY(hipEventCreate(&e->ev0));
    HIP_TRY(hipEventCreate(&e->ev1));
    HIP_TRY(hipMalloc((void **)&e->dm_dev, sizeof(DevModel)));
    HIP_TRY(hipMemcpy(e->dm_dev, &e->dm, sizeof(DevModel),
                      hipMemcpyHostToDevice));
    *out = e;
    return 0;
}

extern "C" void qmc_engine_destroy(qmc_engine *e)
{
    if (!e) return;
    hipSetDevice(e->device);
    if (e->ev0) hipEventDestroy(e->ev0);
    if (e->ev1) hipEventDestroy(e->ev1);
    if (e->dm_dev) hipFree(e->dm_dev);
    if (e->own_stream && e->stream) hipStreamDestroy(e->stream);
    delete e;
}

extern "C" int qmc_engine_sync(qmc_engine *e)
{
    if (!e) return fail("null engine");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

extern "C" int qmc_engine_timer_start(qmc_engine *e)
{
    if (!e) return fail("null engine");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    return 0;
}

extern "C" int qmc_engine_timer_stop(qmc_engine *e, float *ms)
{
    if (!e || !ms) return fail("null argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventRecord(e->ev1, e->stream));
    HIP_TRY(hipEventSynchronize(e->ev1));
    HIP_TRY(hipEventElapsedTime(ms, e->ev0, e->ev1));
    return 0;
}

template <typename T>
static int dev_alloc(T **p, size_t count)
{
    *p = nullptr;
    if (count == 0) count = 1;
    HIP_TRY(hipMalloc((void **)p, count * sizeof(T)));
    return 0;
}

extern "C" int qmc_evaluate_dev(qmc_engine *e, int64_t nconf,
                                const double *pos, double *wf, double *energy,
                                double *ith, double *drift)
{
    if (!e || !pos) return fail("qmc_evaluate_dev: null argument");
    HIP_TRY(hipSetDevice(e->device));
    EvalArgs a{ pos, wf, energy, ith, drift, (long long)nconf };
    return dispatch_shape<LaunchEval>(e, a);
}

// Plain device buffers (configuration sets kept resident across
// qmc_evaluate_dev calls by callers without a device-memory library).
extern "C" int qmc_buffer_alloc(int device, size_t bytes, void **out)
{
    if (!out || !bytes) return fail("qmc_buffer_alloc: null argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMalloc(out, bytes));
    return 0;
}

extern "C" int qmc_buffer_free(void *buf)
{
    if (buf) HIP_TRY(hipFree(buf));
    return 0;
}

extern "C" int qmc_buffer_upload(void *dst_dev, const void *src_host,
                                 size_t bytes)
{
    if (!dst_dev || !src_host) return fail("qmc_buffer_upload: null argument");
    HIP_TRY(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int qmc_buffer_download(void *dst_host, const void *src_dev,
                                   size_t bytes)
{
    if (!dst_host || !src_dev)
        return fail("qmc_buffer_download: null argument");
    HIP_TRY(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int qmc_evaluate(qmc_engine *e, int64_t nconf, const double *pos,
                            double *wf, double *energy, double *ith,
                            double *drift)
{
    if (!e || !pos) return fail("qmc_evaluate: null argument");
    if (nconf <= 0) return 0;
    HIP_TRY(hipSetDevice(e->device));
    const size_t n = (size_t)e->dm.n, W = (size_t)nconf;
    double *dpos, *dwf, *den, *dith, *ddr;
    if (dev_alloc(&dpos, W * n) || dev_alloc(&dwf, W) || dev_alloc(&den, W) ||
        dev_alloc(&dith, W * n) || dev_alloc(&ddr, W * n))
        return 1;
    HIP_TRY(hipMemcpyAsync(dpos, pos, W * n * sizeof(double),
                           hipMemcpyHostToDevice, e->stream));
    int rc = qmc_evaluate_dev(e, nconf, dpos, dwf, den, dith, ddr);
    if (!rc) {
        if (wf) HIP_TRY(hipMemcpyAsync(wf, dwf, W * sizeof(double),
                                       hipMemcpyDeviceToHost, e->stream));
        if (energy) HIP_TRY(hipMemcpyAsync(energy, den, W * sizeof(double),
                                           hipMemcpyDeviceToHost, e->stream));
        if (ith) HIP_TRY(hipMemcpyAsync(ith, dith, W * n * sizeof(double),
                                        hipMemcpyDeviceToHost, e->stream));
        if (drift) HIP_TRY(hipMemcpyAsync(drift, ddr, W * n * sizeof(double),
                                          hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    hipFree(dpos); hipFree(dwf); hipFree(den); hipFree(dith); hipFree(ddr);
    return rc;
}

// Sort every configuration by position; label[w][lane] = original particle
// index held by the lane (see resort_step in qmc_device.h).
static void sort_rows(const double *pos, size_t W, size_t n,
                      std::vector<double> &sorted,
                      std::vector<unsigned short> &label,
                      const double *extra = nullptr,
                      std::vector<double> *extra_sorted = nullptr,
                      size_t pos_stride = 0, size_t extra_stride = 0)
{
    if (!pos_stride) pos_stride = n;
    if (!extra_stride) extra_stride = n;
    sorted.resize(W * n);
    label.resize(W * n);
    if (extra_sorted) extra_sorted->resize(W * n);
    std::vector<unsigned short> idx(n);
    for (size_t w = 0; w < W; ++w) {
        const double *row = pos + w * pos_stride;
        for (size_t i = 0; i < n; ++i) idx[i] = (unsigned short)i;
        std::stable_sort(idx.begin(), idx.end(),
                         [row](unsigned short a, unsigned short b) {
                             return row[a] < row[b];
                         });
        for (size_t i = 0; i < n; ++i) {
            sorted[w * n + i] = row[idx[i]];
            label[w * n + i] = idx[i];
            if (extra_sorted)
                (*extra_sorted)[w * n + i] = extra[w * extra_stride + idx[i]];
        }
    }
}

// ----------------------------------------------------------------- VMC ----
struct qmc_vmc {
    qmc_engine *eng = nullptr;
    qmc_vmc_params p;
    long long W = 0;
    double *pos = nullptr, *wf = nullptr, *ecarry = nullptr;
    unsigned short *label = nullptr;
    double *sum_e = nullptr, *sum_e2 = nullptr;
    long long *n_acc = nullptr;
    double *tape = nullptr;
    long long tape_steps = 0, tape_used = 0;
    unsigned int step = 0;
    int yield_initial = 1;
    double *ssf_partial = nullptr, *ssf_out = nullptr;   // qmc_vmc_ssf scratch
    int ssf_cap = 0;
};

extern "C" int qmc_vmc_create(qmc_engine *e, const qmc_vmc_params *p,
                              qmc_vmc **out)
{
    if (!e || !p || !out) return fail("qmc_vmc_create: null argument");
    if (p->num_chains <= 0) return fail("qmc_vmc_create: num_chains <= 0");
    HIP_TRY(hipSetDevice(e->device));
    qmc_vmc *v = new qmc_vmc();
    v->eng = e;
    v->p = *p;
    v->W = p->num_chains;
    const size_t W = (size_t)v->W, n = (size_t)e->dm.n;
    if (dev_alloc(&v->pos, W * n) || dev_alloc(&v->label, W * n) ||
        dev_alloc(&v->wf, W) || dev_alloc(&v->ecarry, W) || dev_alloc(&v->sum_e, W) ||
        dev_alloc(&v->sum_e2, W) || dev_alloc(&v->n_acc, W)) {
        delete v;
        return 1;
    }
    *out = v;
    return 0;
}

extern "C" void qmc_vmc_destroy(qmc_vmc *v)
{
    if (!v) return;
    hipSetDevice(v->eng->device);
    hipFree(v->pos); hipFree(v->label); hipFree(v->wf); hipFree(v->ecarry);
    if (v->ssf_partial) hipFree(v->ssf_partial);
    if (v->ssf_out) hipFree(v->ssf_out);
    hipFree(v->sum_e); hipFree(v->sum_e2); hipFree(v->n_acc);
    if (v->tape) hipFree(v->tape);
    delete v;
}

extern "C" int qmc_vmc_set_state(qmc_vmc *v, const double *pos)
{
    if (!v || !pos) return fail("qmc_vmc_set_state: null argument");
    qmc_engine *e = v->eng;
    HIP_TRY(hipSetDevice(e->device));
    const size_t W = (size_t)v->W, n = (size_t)e->dm.n;
    std::vector<double> sorted;
    std::vector<unsigned short> label;
    sort_rows(pos, W, n, sorted, label);
    HIP_TRY(hipMemcpyAsync(v->pos, sorted.data(), W * n * sizeof(double),
                           hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(v->label, label.data(), W * n * sizeof(unsigned short),
                           hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemsetAsync(v->ecarry, 0, W * sizeof(double), e->stream));
    int rc = qmc_evaluate_dev(e, v->W, v->pos, v->wf, nullptr, nullptr,
                              nullptr);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    v->step = 0;
    v->yield_initial = 1;
    v->tape_used = 0;
    return 0;
}

extern "C" int qmc_vmc_get_state(qmc_vmc *v, double *pos, double *wf,
                                 double *ecarry)
{
    if (!v) return fail("qmc_vmc_get_state: null argument");
    qmc_engine *e = v->eng;
    HIP_TRY(hipSetDevice(e->device));
    const size_t W = (size_t)v->W, n = (size_t)e->dm.n;
    std::vector<double> lane_pos;
    std::vector<unsigned short> label;
    if (pos) {
        lane_pos.resize(W * n);
        label.resize(W * n);
        HIP_TRY(hipMemcpyAsync(lane_pos.data(), v->pos, W * n * sizeof(double),
                               hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(label.data(), v->label,
                               W * n * sizeof(unsigned short),
                               hipMemcpyDeviceToHost, e->stream));
    }
    if (wf) HIP_TRY(hipMemcpyAsync(wf, v->wf, W * sizeof(double),
                                   hipMemcpyDeviceToHost, e->stream));
    if (ecarry) HIP_TRY(hipMemcpyAsync(ecarry, v->ecarry, W * sizeof(double),
                                       hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (pos)       // back to the caller's particle order
        for (size_t w = 0; w < W; ++w)
            for (size_t i = 0; i < n; ++i)
                pos[w * n + label[w * n + i]] = lane_pos[w * n + i];
    return 0;
}

extern "C" int qmc_vmc_set_tape(qmc_vmc *v, const double *tape, int64_t steps)
{
    if (!v) return fail("qmc_vmc_set_tape: null argument");
    qmc_engine *e = v->eng;
    HIP_TRY(hipSetDevice(e->device));
    if (v->tape) { hipFree(v->tape); v->tape = nullptr; }
    v->tape_steps = 0;
    v->tape_used = 0;
    if (!tape || steps <= 0) return 0;
    size_t cnt = (size_t)v->W * (size_t)steps * (size_t)(e->dm.n + 1);
    if (dev_alloc(&v->tape, cnt)) return 1;
    HIP_TRY(hipMemcpy(v->tape, tape, cnt * sizeof(double),
                      hipMemcpyHostToDevice));
    v->tape_steps = steps;
    return 0;
}

extern "C" int qmc_vmc_state_dev(qmc_vmc *v, double **pos, double **wf)
{
    if (!v) return fail("null argument");
    if (pos) *pos = v->pos;
    if (wf) *wf = v->wf;
    return 0;
}

// Static structure factor parts of the current configurations, summed over
// the chains: out[m] = sum_w (|rho_m|^2, Re rho_m, Im rho_m), m < num_modes
// (qmc_base/jastrow/vmc.py:304-351 evaluates them per step of one chain; an
// ensemble gets them in one launch of the matrix-core kernel).
extern "C" int qmc_vmc_ssf(qmc_vmc *v, int32_t num_modes, double *out)
{
    if (!v || !out) return fail("qmc_vmc_ssf: null argument");
    if (num_modes <= 0 || num_modes > EST_MAXK)
        return fail("qmc_vmc_ssf: num_modes must be in [1, 256]");
    qmc_engine *e = v->eng;
    HIP_TRY(hipSetDevice(e->device));
    const int M = num_modes;
    if (M > v->ssf_cap) {
        if (v->ssf_partial) { hipFree(v->ssf_partial); hipFree(v->ssf_out); }
        v->ssf_partial = v->ssf_out = nullptr;
        if (dev_alloc(&v->ssf_partial, (size_t)EST_BLOCKS * M * 3) ||
            dev_alloc(&v->ssf_out, (size_t)M * 3))
            return 1;
        v->ssf_cap = M;
    }
    EstArgs a;
    a.ppos = v->pos; a.ref = nullptr; a.ctl = nullptr;
    a.aux_prev = nullptr; a.aux_act = nullptr; a.partial = v->ssf_partial;
    a.maxw = v->W; a.step_idx = 0; a.pfw = 0; a.n = e->dm.n; a.K = M;
    a.pure = 0; a.scale = 4.0 / e->dm.L;
    if (M <= 64) {
        const size_t lds = (size_t)(BLOCK / 64) * SsfShape<8>::WAVE_DOUBLES *
                           sizeof(double);
        allow_lds(dmc_ssf_mfma_kernel<8>, lds);
        hipLaunchKernelGGL(dmc_ssf_mfma_kernel<8>, dim3(EST_BLOCKS),
                           dim3(BLOCK), lds, e->stream, a);
    } else {
        const size_t lds = (size_t)(BLOCK / 64) * SsfShape<16>::WAVE_DOUBLES *
                           sizeof(double);
        allow_lds(dmc_ssf_mfma_kernel<16>, lds);
        hipLaunchKernelGGL(dmc_ssf_mfma_kernel<16>, dim3(EST_BLOCKS),
                           dim3(BLOCK), lds, e->stream, a);
    }
    hipLaunchKernelGGL(est_reduce_kernel, dim3((M * 3 + 31) / 32), dim3(256), 0,
                       e->stream, v->ssf_partial, EST_BLOCKS, M * 3, 1.0,
                       v->ssf_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, v->ssf_out, (size_t)M * 3 * sizeof(double),
                           hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

extern "C" int qmc_vmc_block_sums_dev(qmc_vmc *v, double **se, double **se2,
                                      int64_t **na)
{
    if (!v) return fail("null argument");
    if (se) *se = v->sum_e;
    if (se2) *se2 = v->sum_e2;
    if (na) *na = (int64_t *)v->n_acc;
    return 0;
}

extern "C" int qmc_vmc_run_block(qmc_vmc *v, int64_t nyield, double *sum_e,
                                 double *sum_e2, int64_t *n_acc,
                                 double *ser_wf, double *ser_e,
                                 uint8_t *ser_stat, double *ser_pos)
{
    if (!v) return fail("qmc_vmc_run_block: null argument");
    if (nyield <= 0) return fail("qmc_vmc_run_block: nyield must be >= 1");
    qmc_engine *e = v->eng;
    HIP_TRY(hipSetDevice(e->device));
    const size_t W = (size_t)v->W, ny = (size_t)nyield;
    const long long real = nyield - (v->yield_initial ? 1 : 0);
    if (v->tape && v->tape_used + real > v->tape_steps)
        return fail("qmc_vmc_run_block: tape exhausted");
    double *dwf = nullptr, *de = nullptr;
    unsigned char *dst = nullptr;
    double *dpos = nullptr;
    if (ser_pos && dev_alloc(&dpos, ny * W * (size_t)e->dm.n)) return 1;
    if (ser_wf && dev_alloc(&dwf, ny * W)) return 1;
    if (ser_e && dev_alloc(&de, ny * W)) return 1;
    if (ser_stat && dev_alloc(&dst, ny * W)) return 1;
    VmcArgs a;
    a.pos = v->pos; a.label = v->label; a.wf = v->wf; a.ecarry = v->ecarry;
    a.sum_e = v->sum_e; a.sum_e2 = v->sum_e2; a.n_acc = v->n_acc;
    a.ser_wf = dwf; a.ser_e = de; a.ser_stat = dst; a.ser_pos = dpos;
    a.tape = v->tape;
    a.tape_steps = v->tape_steps;
    a.W = v->W;
    a.gaussian = v->p.gaussian;
    a.chain0 = v->p.chain0;
    a.seed = v->p.rng_seed; a.move_spread = v->p.move_spread;
    for (long long y = 0; y < nyield; ++y) {
        a.y = y;
        a.forced = (y == 0 && v->yield_initial) ? 1 : 0;
        a.reset_sums = (y == 0) ? 1 : 0;
        a.step = v->step;
        a.tape_idx = v->tape_used;
        int rc = dispatch_shape<LaunchVmc>(e, a);
        if (rc) return rc;
        if (!a.forced) { v->step += 1; v->tape_used += 1; }
    }
    v->yield_initial = 0;
    bool need_sync = false;
    if (sum_e) { HIP_TRY(hipMemcpyAsync(sum_e, v->sum_e, W * sizeof(double),
                 hipMemcpyDeviceToHost, e->stream)); need_sync = true; }
    if (sum_e2) { HIP_TRY(hipMemcpyAsync(sum_e2, v->sum_e2, W * sizeof(double),
                  hipMemcpyDeviceToHost, e->stream)); need_sync = true; }
    if (n_acc) { HIP_TRY(hipMemcpyAsync(n_acc, v->n_acc, W * sizeof(int64_t),
                 hipMemcpyDeviceToHost, e->stream)); need_sync = true; }
    if (ser_wf) { HIP_TRY(hipMemcpyAsync(ser_wf, dwf, ny * W * sizeof(double),
                  hipMemcpyDeviceToHost, e->stream)); need_sync = true; }
    if (ser_e) { HIP_TRY(hipMemcpyAsync(ser_e, de, ny * W * sizeof(double),
                 hipMemcpyDeviceToHost, e->stream)); need_sync = true; }
    if (ser_stat) { HIP_TRY(hipMemcpyAsync(ser_stat, dst, ny * W,
                    hipMemcpyDeviceToHost, e->stream)); need_sync = true; }
    if (ser_pos) { HIP_TRY(hipMemcpyAsync(ser_pos, dpos,
                   ny * W * (size_t)e->dm.n * sizeof(double),
                   hipMemcpyDeviceToHost, e->stream)); need_sync = true; }
    if (need_sync) HIP_TRY(hipStreamSynchronize(e->stream));
    if (dpos) hipFree(dpos);
    if (dwf) hipFree(dwf);
    if (de) hipFree(de);
    if (dst) hipFree(dst);
    return 0;
}

// ----------------------------------------------------------------- DMC ----
struct qmc_dmc {
    qmc_engine *eng = nullptr;
    qmc_dmc_params p;
    long long maxw = 0;
    int nblocks = 0;
    // two population buffers: [0]/[1] alternate parent / child roles
    double *pos[2] = { nullptr, nullptr }, *drift[2] = { nullptr, nullptr };
    unsigned short *label[2] = { nullptr, nullptr };
    double *energy[2] = { nullptr, nullptr }, *weight[2] = { nullptr, nullptr };
    int cur = 0;                 // index of the parent buffer
    double *eslot = nullptr;
    double *spare = nullptr;
    long long *ref = nullptr;
    int *count = nullptr;
    long long *block_tot = nullptr, *block_off = nullptr;
    double *block_esum = nullptr;
    DmcCtl *ctl = nullptr;
    // per-step series (device), capacity ser_cap steps
    double *ser_e = nullptr, *ser_w = nullptr, *ser_ref = nullptr,
           *ser_acc = nullptr;
    unsigned long long *ser_nw = nullptr;
    long long ser_cap = 0, ser_len = 0;
    // tapes (test only)
    double *u_tape = nullptr, *g_tape = nullptr;
    std::vector<long long> u_off, g_off;
    long long tape_step = 0;
    bool stepped = false;        // a step has run since the last set_state
    bool sums_pending = false;   // E_t partials still to be summed (by finish)
    double global_target = 0.0;
    // estimators (f1)
    qmc_dmc_est_params est;
    bool have_est = false;
    double *ssf_aux[2] = { nullptr, nullptr };   // [maxw][M][3]
    double *dens_aux[2] = { nullptr, nullptr };  // [maxw][B]
    double *est_partial = nullptr;               // [EST_BLOCKS][max(3M, B)]
    double *iter_ssf = nullptr, *iter_dens = nullptr;
    long long iter_cap = 0;
};

static int dmc_reserve_series(qmc_dmc *d, long long nsteps)
{
    if (nsteps <= d->ser_cap) return 0;
    if (d->ser_e) { hipFree(d->ser_e); hipFree(d->ser_w); hipFree(d->ser_ref);
                    hipFree(d->ser_acc); hipFree(d->ser_nw); }
    if (dev_alloc(&d->ser_e, nsteps) || dev_alloc(&d->ser_w, nsteps) ||
        dev_alloc(&d->ser_ref, nsteps) || dev_alloc(&d->ser_acc, nsteps) ||
        dev_alloc(&d->ser_nw, nsteps))
        return 1;
    d->ser_cap = nsteps;
    return 0;
}

extern "C" int qmc_dmc_create(qmc_engine *e, const qmc_dmc_params *p,
                              qmc_dmc **out)
{
    if (!e || !p || !out) return fail("qmc_dmc_create: null argument");
    if (p->max_num_walkers <= 0 || p->target_num_walkers <= 0)
        return fail("qmc_dmc_create: walker counts must be positive");
    if (!(p->time_step > 0)) return fail("qmc_dmc_create: time_step <= 0");
    HIP_TRY(hipSetDevice(e->device));
    qmc_dmc *d = new qmc_dmc();
    d->eng = e;
    d->p = *p;
    d->maxw = p->max_num_walkers;
    d->nblocks = (int)((d->maxw + BR_TILE - 1) / BR_TILE);
    d->global_target = (double)p->target_num_walkers;
    const size_t W = (size_t)d->maxw, n = (size_t)e->dm.n;
    int rc = 0;
    for (int b = 0; b < 2 && !rc; ++b) {
        rc |= dev_alloc(&d->pos[b], W * n) || dev_alloc(&d->drift[b], W * n) ||
              dev_alloc(&d->label[b], W * n) ||
              dev_alloc(&d->energy[b], W) || dev_alloc(&d->weight[b], W);
    }
    rc = rc || dev_alloc(&d->eslot, W) || dev_alloc(&d->spare, W * n) ||
         dev_alloc(&d->ref, W) ||
         dev_alloc(&d->count, W) || dev_alloc(&d->block_tot, d->nblocks) ||
         dev_alloc(&d->block_off, d->nblocks) ||
         dev_alloc(&d->block_esum, d->nblocks) || dev_alloc(&d->ctl, 1);
    if (rc) { delete d; return 1; }
    HIP_TRY(hipMemset(d->ctl, 0, sizeof(DmcCtl)));
    HIP_TRY(hipMemset(d->ref, 0, W * sizeof(long long)));
    HIP_TRY(hipMemset(d->eslot, 0, W * sizeof(double)));
    *out = d;
    return 0;
}

extern "C" void qmc_dmc_destroy(qmc_dmc *d)
{
    if (!d) return;
    hipSetDevice(d->eng->device);
    for (int b = 0; b < 2; ++b) {
        hipFree(d->pos[b]); hipFree(d->drift[b]); hipFree(d->label[b]);
        hipFree(d->energy[b]); hipFree(d->weight[b]);
    }
    hipFree(d->eslot); hipFree(d->spare); hipFree(d->ref); hipFree(d->count);
    hipFree(d->block_tot); hipFree(d->block_off); hipFree(d->block_esum);
    hipFree(d->ctl);
    if (d->ser_e) { hipFree(d->ser_e); hipFree(d->ser_w); hipFree(d->ser_ref);
                    hipFree(d->ser_acc); hipFree(d->ser_nw); }
    if (d->u_tape) hipFree(d->u_tape);
    if (d->g_tape) hipFree(d->g_tape);
    for (int k = 0; k < 2; ++k) {
        if (d->ssf_aux[k]) hipFree(d->ssf_aux[k]);
        if (d->dens_aux[k]) hipFree(d->dens_aux[k]);
    }
    if (d->est_partial) hipFree(d->est_partial);
    if (d->iter_ssf) hipFree(d->iter_ssf);
    if (d->iter_dens) hipFree(d->iter_dens);
    delete d;
}

static int dmc_reset_ctl(qmc_dmc *d, long long nw, double ref_energy)
{
    DmcCtl c;
    memset(&c, 0, sizeof(c));
    c.prev_nw = nw;
    c.nw = nw;
    c.ref_energy = ref_energy;
    HIP_TRY(hipMemcpyAsync(d->ctl, &c, sizeof(c), hipMemcpyHostToDevice,
                           d->eng->stream));
    HIP_TRY(hipStreamSynchronize(d->eng->stream));
    d->cur = 0;
    d->stepped = false;
    d->tape_step = 0;
    d->ser_len = 0;
    return 0;
}

static int dmc_zero_population(qmc_dmc *d)
{
    qmc_engine *e = d->eng;
    const size_t n = (size_t)e->dm.n, W = (size_t)d->maxw;
    std::vector<unsigned short> ident(W * n);
    for (size_t s = 0; s < W; ++s)
        for (size_t i = 0; i < n; ++i) ident[s * n + i] = (unsigned short)i;
    for (int b = 0; b < 2; ++b) {
        HIP_TRY(hipMemsetAsync(d->pos[b], 0, W * n * sizeof(double), e->stream));
        HIP_TRY(hipMemsetAsync(d->drift[b], 0, W * n * sizeof(double), e->stream));
        HIP_TRY(hipMemsetAsync(d->energy[b], 0, W * sizeof(double), e->stream));
        HIP_TRY(hipMemsetAsync(d->weight[b], 0, W * sizeof(double), e->stream));
        HIP_TRY(hipMemcpyAsync(d->label[b], ident.data(),
                               W * n * sizeof(unsigned short),
                               hipMemcpyHostToDevice, e->stream));
    }
    HIP_TRY(hipMemsetAsync(d->eslot, 0, W * sizeof(double), e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

// pos: host positions (sorted here) or device positions already in lane order
// with their labels (label_dev, may be null = identity).
static int dmc_set_state_impl(qmc_dmc *d, int64_t nw, const double *pos,
                              bool pos_on_device,
                              const unsigned short *label_dev, int use_ref,
                              double ref_energy)
{
    if (!d || !pos) return fail("qmc_dmc_set_state: null argument");
    if (nw <= 0 || nw > d->maxw)
        return fail("qmc_dmc_set_state: number of walkers out of range");
    qmc_engine *e = d->eng;
    HIP_TRY(hipSetDevice(e->device));
    const size_t n = (size_t)e->dm.n;
    if (dmc_zero_population(d)) return 1;
    if (pos_on_device) {
        HIP_TRY(hipMemcpyAsync(d->pos[0], pos, (size_t)nw * n * sizeof(double),
                               hipMemcpyDeviceToDevice, e->stream));
        if (label_dev)
            HIP_TRY(hipMemcpyAsync(d->label[0], label_dev,
                                   (size_t)nw * n * sizeof(unsigned short),
                                   hipMemcpyDeviceToDevice, e->stream));
    } else {
        std::vector<double> sorted;
        std::vector<unsigned short> label;
        sort_rows(pos, (size_t)nw, n, sorted, label);
        HIP_TRY(hipMemcpyAsync(d->pos[0], sorted.data(),
                               (size_t)nw * n * sizeof(double),
                               hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(d->label[0], label.data(),
                               (size_t)nw * n * sizeof(unsigned short),
                               hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    PrepArgs a{ d->pos[0], d->drift[0], d->energy[0], (long long)nw };
    int rc = dispatch_shape<LaunchPrep>(e, a);
    if (rc) return rc;
    std::vector<double> ones((size_t)nw, 1.0), en((size_t)nw);
    HIP_TRY(hipMemcpyAsync(d->weight[0], ones.data(), (size_t)nw * sizeof(double),
                           hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(en.data(), d->energy[0], (size_t)nw * sizeof(double),
                           hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(d->eslot, d->energy[0], (size_t)nw * sizeof(double),
                           hipMemcpyDeviceToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (!use_ref) {
        // mrbp_qmc/dmc.py:302-312: sum(E w) / sum(w) with unit weights
        double se = 0.0;
        for (size_t i = 0; i < (size_t)nw; ++i) se += en[i] * 1.0;
        ref_energy = se / (double)nw;
    }
    return dmc_reset_ctl(d, nw, ref_energy);
}

extern "C" int qmc_dmc_set_state(qmc_dmc *d, int64_t nw, const double *pos,
                                 int use_ref, double ref_energy)
{
    return dmc_set_state_impl(d, nw, pos, false, nullptr, use_ref, ref_energy);
}

extern "C" int qmc_dmc_set_state_dev(qmc_dmc *d, int64_t nw,
                                     const double *pos_dev, int use_ref,
                                     double ref_energy)
{
    return dmc_set_state_impl(d, nw, pos_dev, true, nullptr, use_ref,
                              ref_energy);
}

extern "C" int qmc_dmc_set_state_from_vmc(qmc_dmc *d, qmc_vmc *v, int64_t nw,
                                          int use_ref, double ref_energy)
{
    if (!d || !v) return fail("qmc_dmc_set_state_from_vmc: null argument");
    if (v->eng != d->eng)
        return fail("qmc_dmc_set_state_from_vmc: ensembles of different engines");
    if (nw > v->W) return fail("qmc_dmc_set_state_from_vmc: nw > num_chains");
    return dmc_set_state_impl(d, nw, v->pos, true, v->label, use_ref,
                              ref_energy);
}

extern "C" int qmc_dmc_set_full_state(qmc_dmc *d, int64_t nw,
                                      const double *confs,
                                      const double *energy,
                                      const double *weight,
                                      const double *slot_energy,
                                      double ref_energy)
{
    if (!d || !confs || !energy || !weight)
        return fail("qmc_dmc_set_full_state: null argument");
    if (nw <= 0 || nw > d->maxw)
        return fail("qmc_dmc_set_full_state: number of walkers out of range");
    qmc_engine *e = d->eng;
    HIP_TRY(hipSetDevice(e->device));
    const size_t n = (size_t)e->dm.n, W = (size_t)d->maxw;
    // confs[s] = (pos row, drift row): sort by position, carry the drift along
    std::vector<double> hp, hd;
    std::vector<unsigned short> label;
    sort_rows(confs, (size_t)nw, n, hp, label, confs + n, &hd, 2 * n, 2 * n);
    if (dmc_zero_population(d)) return 1;
    HIP_TRY(hipMemcpyAsync(d->pos[0], hp.data(), hp.size() * sizeof(double),
                           hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(d->drift[0], hd.data(), hd.size() * sizeof(double),
                           hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(d->label[0], label.data(),
                           label.size() * sizeof(unsigned short),
                           hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(d->energy[0], energy, (size_t)nw * sizeof(double),
                           hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(d->weight[0], weight, (size_t)nw * sizeof(double),
                           hipMemcpyHostToDevice, e->stream));
    // the reference copies the whole props.energy array of the initial state
    // into its `actual` buffer (qmc_base/dmc.py:707-708), stale tail included
    if (slot_energy)
        HIP_TRY(hipMemcpyAsync(d->eslot, slot_energy, W * sizeof(double),
                               hipMemcpyHostToDevice, e->stream));
    else
        HIP_TRY(hipMemcpyAsync(d->eslot, energy, (size_t)nw * sizeof(double),
                               hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return dmc_reset_ctl(d, nw, ref_energy);
}

extern "C" int qmc_dmc_set_tape(qmc_dmc *d, const double *u, int64_t nu,
                                const double *g, int64_t ng,
                                const int64_t *u_off, const int64_t *g_off,
                                int64_t nsteps)
{
    if (!d) return fail("qmc_dmc_set_tape: null argument");
    HIP_TRY(hipSetDevice(d->eng->device));
    if (d->u_tape) { hipFree(d->u_tape); d->u_tape = nullptr; }
    if (d->g_tape) { hipFree(d->g_tape); d->g_tape = nullptr; }
    d->u_off.clear(); d->g_off.clear();
    d->tape_step = 0;
    if (!u || !g || nsteps <= 0) return 0;
    // pad so that a step may read up to maxw uniforms / maxw*N normals
    size_t upad = (size_t)nu + (size_t)d->maxw;
    size_t gpad = (size_t)ng + (size_t)d->maxw * (size_t)d->eng->dm.n;
    if (dev_alloc(&d->u_tape, upad) || dev_alloc(&d->g_tape, gpad)) return 1;
    HIP_TRY(hipMemset(d->u_tape, 0, upad * sizeof(double)));
    HIP_TRY(hipMemset(d->g_tape, 0, gpad * sizeof(double)));
    HIP_TRY(hipMemcpy(d->u_tape, u, (size_t)nu * sizeof(double),
                      hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d->g_tape, g, (size_t)ng * sizeof(double),
                      hipMemcpyHostToDevice));
    d->u_off.assign(u_off, u_off + nsteps);
    d->g_off.assign(g_off, g_off + nsteps);
    return 0;
}

// Enqueue the rank-local part of one time step.
static int dmc_enqueue_local(qmc_dmc *d, double *partial_dev)
{
    qmc_engine *e = d->eng;
    const int par = d->cur, chi = 1 - d->cur;
    const double *ut = nullptr, *gt = nullptr;
    if (d->u_tape) {
        if (d->tape_step >= (long long)d->u_off.size())
            return fail("qmc_dmc: tape exhausted");
        ut = d->u_tape + d->u_off[(size_t)d->tape_step];
        gt = d->g_tape + d->g_off[(size_t)d->tape_step];
        d->tape_step += 1;
    }
    BranchArgs b;
    b.weight = d->weight[par]; b.energy = d->energy[par];
    b.count = d->count; b.block_tot = d->block_tot; b.block_off = d->block_off;
    b.block_esum = d->block_esum; b.ref = d->ref; b.ctl = d->ctl;
    b.u_tape = ut; b.maxw = d->maxw; b.seed = d->p.rng_seed;
    b.slot0 = d->p.slot0;
    if (d->nblocks <= BR_FUSED_TILES) {
        d->sums_pending = false;
        hipLaunchKernelGGL(branch_fused_kernel, dim3(1), dim3(BLOCK), 0,
                           e->stream, b, partial_dev);
    } else {
        hipLaunchKernelGGL(branch_count_kernel, dim3(d->nblocks), dim3(BLOCK),
                           0, e->stream, b);
        hipLaunchKernelGGL(branch_scatter_kernel, dim3(d->nblocks),
                           dim3(BLOCK), 0, e->stream, b);
        // single GPU: the finish kernel sums the partials itself
        d->sums_pending = partial_dev == nullptr;
        if (partial_dev)
            hipLaunchKernelGGL(dmc_local_sums_kernel, dim3(1), dim3(BLOCK), 0,
                               e->stream, d->block_esum, d->ctl, partial_dev);
    }
    HIP_TRY(hipGetLastError());
    EvolveArgs a;
    a.ppos = d->pos[par]; a.pdrift = d->drift[par]; a.penergy = d->energy[par];
    a.cpos = d->pos[chi]; a.cdrift = d->drift[chi];
    a.plabel = d->label[par]; a.clabel = d->label[chi];
    a.cenergy = d->energy[chi]; a.cweight = d->weight[chi];
    a.eslot = d->eslot; a.ref = d->ref; a.ctl = d->ctl; a.g_tape = gt;
    a.spare = d->spare;
    a.maxw = d->maxw; a.dt = d->p.time_step;
    a.sigma = sqrt(2 * d->p.time_step);           // mrbp_qmc/dmc.py:178
    a.seed = d->p.rng_seed; a.slot0 = d->p.slot0;
    a.fix_stale = d->p.fix_stale_energy;
    int rc = dispatch_shape<LaunchEvolve>(e, a);
    if (rc) return rc;
    return 0;
}

static int dmc_enqueue_finish(qmc_dmc *d, const double *total_dev,
                              long long ser_idx)
{
    qmc_engine *e = d->eng;
    FinishArgs f;
    f.ctl = d->ctl; f.total = total_dev;
    f.block_esum = (!total_dev && d->sums_pending) ? d->block_esum : nullptr;
    const bool rec = ser_idx >= 0;
    f.ser_e = rec ? d->ser_e : nullptr; f.ser_w = d->ser_w;
    f.ser_ref = d->ser_ref; f.ser_acc = d->ser_acc; f.ser_nw = d->ser_nw;
    f.ser_idx = ser_idx;
    f.kappa = d->p.num_walkers_control_factor; f.dt = d->p.time_step;
    f.target = d->global_target;
    hipLaunchKernelGGL(dmc_finish_kernel, dim3(1), dim3(BLOCK), 0, e->stream,
                       f);
    HIP_TRY(hipGetLastError());
    d->cur = 1 - d->cur;          // children become the parents
    d->stepped = true;
    return 0;
}

extern "C" int qmc_dmc_read_series(qmc_dmc *d, int64_t nsteps, double *energy,
                                   double *weight, uint64_t *num_walkers,
                                   double *ref_energy, double *accum_energy)
{
    if (!d) return fail("qmc_dmc_read_series: null argument");
    if (nsteps > d->ser_len) return fail("qmc_dmc_read_series: too many steps");
    qmc_engine *e = d->eng;
    HIP_TRY(hipSetDevice(e->device));
    const size_t ns = (size_t)nsteps;
    if (energy) HIP_TRY(hipMemcpyAsync(energy, d->ser_e, ns * 8,
                                       hipMemcpyDeviceToHost, e->stream));
    if (weight) HIP_TRY(hipMemcpyAsync(weight, d->ser_w, ns * 8,
                                       hipMemcpyDeviceToHost, e->stream));
    if (num_walkers) HIP_TRY(hipMemcpyAsync(num_walkers, d->ser_nw, ns * 8,
                                            hipMemcpyDeviceToHost, e->stream));
    if (ref_energy) HIP_TRY(hipMemcpyAsync(ref_energy, d->ser_ref, ns * 8,
                                           hipMemcpyDeviceToHost, e->stream));
    if (accum_energy) HIP_TRY(hipMemcpyAsync(accum_energy, d->ser_acc, ns * 8,
                                             hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    // a full read consumes the series (split-step drivers append to it)
    if (nsteps == d->ser_len) d->ser_len = 0;
    return 0;
}

extern "C" int qmc_dmc_run_block(qmc_dmc *d, int64_t nsteps, double *energy,
                                 double *weight, uint64_t *num_walkers,
                                 double *ref_energy, double *accum_energy)
{
    if (!d) return fail("qmc_dmc_run_block: null argument");
    if (nsteps <= 0) return fail("qmc_dmc_run_block: nsteps must be >= 1");
    if (d->p.external_reduce)
        return fail("qmc_dmc_run_block: ensemble was created for "
                    "external_reduce; drive it with step_local/step_finish");
    HIP_TRY(hipSetDevice(d->eng->device));
    if (dmc_reserve_series(d, nsteps)) return 1;
    d->ser_len = 0;
    for (long long t = 0; t < nsteps; ++t) {
        int rc = dmc_enqueue_local(d, nullptr);
        if (rc) return rc;
        rc = dmc_enqueue_finish(d, nullptr, t);
        if (rc) return rc;
        d->ser_len = t + 1;
    }
    if (energy || weight || num_walkers || ref_energy || accum_energy)
        return qmc_dmc_read_series(d, nsteps, energy, weight, num_walkers,
                                   ref_energy, accum_energy);
    return 0;
}

extern "C" int qmc_dmc_set_estimators(qmc_dmc *d, const qmc_dmc_est_params *p)
{
    if (!d || !p) return fail("qmc_dmc_set_estimators: null argument");
    if (p->num_modes < 0 || p->num_modes > EST_MAXK || p->num_bins < 0 ||
        p->num_bins > EST_MAXK)
        return fail("qmc_dmc_set_estimators: num_modes / num_bins must be in "
                    "[0, 256]");
    qmc_engine *e = d->eng;
    HIP_TRY(hipSetDevice(e->device));
    for (int k = 0; k < 2; ++k) {
        if (d->ssf_aux[k]) { hipFree(d->ssf_aux[k]); d->ssf_aux[k] = nullptr; }
        if (d->dens_aux[k]) { hipFree(d->dens_aux[k]); d->dens_aux[k] = nullptr; }
    }
    if (d->est_partial) { hipFree(d->est_partial); d->est_partial = nullptr; }
    d->est = *p;
    d->have_est = p->num_modes > 0 || p->num_bins > 0;
    const size_t W = (size_t)d->maxw;
    if (p->num_modes > 0)
        for (int k = 0; k < 2; ++k)
            if (dev_alloc(&d->ssf_aux[k], W * (size_t)p->num_modes * 3)) return 1;
    if (p->num_bins > 0)
        for (int k = 0; k < 2; ++k)
            if (dev_alloc(&d->dens_aux[k], W * (size_t)p->num_bins)) return 1;
    size_t kc = (size_t)(p->num_modes * 3 > p->num_bins ? p->num_modes * 3
                                                        : p->num_bins);
    if (d->have_est && dev_alloc(&d->est_partial, EST_BLOCKS * kc)) return 1;
    return 0;
}

// Evaluate the estimators on the population yielded by the step that has just
// been finished (its parents are the buffer that is not `cur`).
static int dmc_enqueue_estimators(qmc_dmc *d, long long step_idx)
{
    qmc_engine *e = d->eng;
    const int par = 1 - d->cur;
    const int act = (int)(step_idx % 2), prev = 1 - act;
    EstArgs a;
    a.ppos = d->pos[par]; a.ref = d->ref; a.ctl = d->ctl;
    a.maxw = d->maxw; a.step_idx = step_idx; a.n = e->dm.n;
    a.partial = d->est_partial;
    if (d->est.num_modes > 0) {
        const int M = d->est.num_modes;
        a.aux_prev = d->ssf_aux[prev]; a.aux_act = d->ssf_aux[act];
        a.K = M; a.pure = d->est.ssf_pure; a.pfw = d->est.ssf_pfw;
        a.scale = 4.0 / e->dm.L;
        if (M <= 64) {
            const size_t lds = (size_t)(BLOCK / 64) *
                               SsfShape<8>::WAVE_DOUBLES * sizeof(double);
            allow_lds(dmc_ssf_mfma_kernel<8>, lds);
            hipLaunchKernelGGL(dmc_ssf_mfma_kernel<8>, dim3(EST_BLOCKS),
                               dim3(BLOCK), lds, e->stream, a);
        } else {
            const size_t lds = (size_t)(BLOCK / 64) *
                               SsfShape<16>::WAVE_DOUBLES * sizeof(double);
            allow_lds(dmc_ssf_mfma_kernel<16>, lds);
            hipLaunchKernelGGL(dmc_ssf_mfma_kernel<16>, dim3(EST_BLOCKS),
                               dim3(BLOCK), lds, e->stream, a);
        }
        double div = 1.0;
        if (a.pure) div = step_idx < a.pfw ? (double)(step_idx + 1)
                                           : (double)a.pfw;
        hipLaunchKernelGGL(est_reduce_kernel, dim3((M * 3 + 31) / 32),
                           dim3(256), 0, e->stream, d->est_partial, EST_BLOCKS,
                           M * 3, div, d->iter_ssf + (size_t)step_idx * M * 3);
    }
    if (d->est.num_bins > 0) {
        const int B = d->est.num_bins;
        a.aux_prev = d->dens_aux[prev]; a.aux_act = d->dens_aux[act];
        a.K = B; a.pure = d->est.dens_pure; a.pfw = d->est.dens_pfw;
        a.scale = e->dm.L / (double)B;      // bin size
        hipLaunchKernelGGL(dmc_density_kernel, dim3(EST_BLOCKS), dim3(BLOCK), 0,
                           e->stream, a);
        double div = 1.0;
        if (a.pure) div = step_idx < a.pfw ? (double)(step_idx + 1)
                                           : (double)a.pfw;
        hipLaunchKernelGGL(est_reduce_kernel, dim3((B + 31) / 32), dim3(256),
                           0, e->stream, d->est_partial, EST_BLOCKS, B, div,
                           d->iter_dens + (size_t)step_idx * B);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int qmc_dmc_run_block_est(qmc_dmc *d, int64_t nsteps,
                                     int eval_estimators, double *energy,
                                     double *weight, uint64_t *num_walkers,
                                     double *ref_energy, double *accum_energy,
                                     double *iter_ssf, double *iter_density)
{
    if (!d) return fail("qmc_dmc_run_block_est: null argument");
    if (nsteps <= 0) return fail("qmc_dmc_run_block_est: nsteps must be >= 1");
    if (d->p.external_reduce)
        return fail("qmc_dmc_run_block_est: not available with external_reduce");
    if (!d->have_est)
        return qmc_dmc_run_block(d, nsteps, energy, weight, num_walkers,
                                 ref_energy, accum_energy);
    qmc_engine *e = d->eng;
    HIP_TRY(hipSetDevice(e->device));
    if (dmc_reserve_series(d, nsteps)) return 1;
    const size_t M3 = (size_t)d->est.num_modes * 3, B = (size_t)d->est.num_bins;
    const size_t W = (size_t)d->maxw;
    if (nsteps > d->iter_cap) {
        if (d->iter_ssf) { hipFree(d->iter_ssf); d->iter_ssf = nullptr; }
        if (d->iter_dens) { hipFree(d->iter_dens); d->iter_dens = nullptr; }
        if (dev_alloc(&d->iter_ssf, (size_t)nsteps * (M3 ? M3 : 1)) ||
            dev_alloc(&d->iter_dens, (size_t)nsteps * (B ? B : 1)))
            return 1;
        d->iter_cap = nsteps;
    }
    // per-block resets (qmc_base/dmc.py:897-909)
    HIP_TRY(hipMemsetAsync(d->iter_ssf, 0, (size_t)nsteps * (M3 ? M3 : 1) * 8,
                           e->stream));
    HIP_TRY(hipMemsetAsync(d->iter_dens, 0, (size_t)nsteps * (B ? B : 1) * 8,
                           e->stream));
    for (int k = 0; k < 2; ++k) {
        if (M3) HIP_TRY(hipMemsetAsync(d->ssf_aux[k], 0, W * M3 * 8, e->stream));
        if (B) HIP_TRY(hipMemsetAsync(d->dens_aux[k], 0, W * B * 8, e->stream));
    }
    d->ser_len = 0;
    for (long long t = 0; t < nsteps; ++t) {
        int rc = dmc_enqueue_local(d, nullptr);
        if (rc) return rc;
        rc = dmc_enqueue_finish(d, nullptr, t);
        if (rc) return rc;
        d->ser_len = t + 1;
        if (eval_estimators) {
            rc = dmc_enqueue_estimators(d, t);
            if (rc) return rc;
        }
    }
    if (iter_ssf && M3)
        HIP_TRY(hipMemcpyAsync(iter_ssf, d->iter_ssf, (size_t)nsteps * M3 * 8,
                               hipMemcpyDeviceToHost, e->stream));
    if (iter_density && B)
        HIP_TRY(hipMemcpyAsync(iter_density, d->iter_dens, (size_t)nsteps * B * 8,
                               hipMemcpyDeviceToHost, e->stream));
    return qmc_dmc_read_series(d, nsteps, energy, weight, num_walkers,
                               ref_energy, accum_energy);
}

extern "C" int qmc_dmc_step_local(qmc_dmc *d, double *partial_dev)
{
    if (!d || !partial_dev) return fail("qmc_dmc_step_local: null argument");
    HIP_TRY(hipSetDevice(d->eng->device));
    return dmc_enqueue_local(d, partial_dev);
}

extern "C" int qmc_dmc_step_finish(qmc_dmc *d, const double *total_dev)
{
    if (!d || !total_dev) return fail("qmc_dmc_step_finish: null argument");
    HIP_TRY(hipSetDevice(d->eng->device));
    if (d->ser_len >= d->ser_cap) {
        // grow geometrically, keeping what was recorded
        long long ncap = d->ser_cap ? d->ser_cap * 2 : 1024;
        double *oe = d->ser_e, *ow = d->ser_w, *orf = d->ser_ref,
               *oa = d->ser_acc;
        unsigned long long *on = d->ser_nw;
        long long olen = d->ser_len;
        d->ser_e = nullptr; d->ser_cap = 0;
        if (dmc_reserve_series(d, ncap)) return 1;
        if (oe) {
            hipStream_t s = d->eng->stream;
            HIP_TRY(hipMemcpyAsync(d->ser_e, oe, olen * 8, hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipMemcpyAsync(d->ser_w, ow, olen * 8, hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipMemcpyAsync(d->ser_ref, orf, olen * 8, hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipMemcpyAsync(d->ser_acc, oa, olen * 8, hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipMemcpyAsync(d->ser_nw, on, olen * 8, hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipStreamSynchronize(s));
            hipFree(oe); hipFree(ow); hipFree(orf); hipFree(oa); hipFree(on);
        }
    }
    int rc = dmc_enqueue_finish(d, total_dev, d->ser_len);
    if (rc) return rc;
    d->ser_len += 1;
    return 0;
}

extern "C" int qmc_dmc_num_walkers(qmc_dmc *d, int64_t *nw)
{
    if (!d || !nw) return fail("qmc_dmc_num_walkers: null argument");
    HIP_TRY(hipSetDevice(d->eng->device));
    DmcCtl c;
    HIP_TRY(hipMemcpyAsync(&c, d->ctl, sizeof(c), hipMemcpyDeviceToHost,
                           d->eng->stream));
    HIP_TRY(hipStreamSynchronize(d->eng->stream));
    *nw = c.prev_nw;
    return 0;
}

extern "C" int qmc_dmc_get_state(qmc_dmc *d, double *confs, double *energy,
                                 double *weight, uint8_t *mask,
                                 int64_t *cloning_ref, double *scalars)
{
    if (!d) return fail("qmc_dmc_get_state: null argument");
    qmc_engine *e = d->eng;
    HIP_TRY(hipSetDevice(e->device));
    const size_t n = (size_t)e->dm.n, W = (size_t)d->maxw;
    DmcCtl c;
    HIP_TRY(hipMemcpyAsync(&c, d->ctl, sizeof(c), hipMemcpyDeviceToHost,
                           e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    const long long nw = d->stepped ? c.nw : c.prev_nw;
    std::vector<long long> href(W, 0);
    if (d->stepped)
        HIP_TRY(hipMemcpy(href.data(), d->ref, W * sizeof(long long),
                          hipMemcpyDeviceToHost));
    if (confs) {
        memset(confs, 0, W * 2 * n * sizeof(double));
        if (d->stepped) {
            // yielded walkers are the parents selected by the cloning table;
            // the parent buffer of the last step is the one that is not `cur`
            const int par = 1 - d->cur;
            double *tmp;
            if (dev_alloc(&tmp, (size_t)nw * 2 * n)) return 1;
            long long tot = nw * (long long)n;
            hipLaunchKernelGGL(dmc_gather_state_kernel,
                               dim3((unsigned)((tot + 255) / 256)), dim3(256),
                               0, e->stream, d->pos[par], d->drift[par],
                               d->label[par], d->ref, nw, (int)n, tmp);
            HIP_TRY(hipMemcpyAsync(confs, tmp, (size_t)nw * 2 * n * 8,
                                   hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
            hipFree(tmp);
        } else {
            std::vector<double> hp((size_t)nw * n), hd((size_t)nw * n);
            std::vector<unsigned short> hl((size_t)nw * n);
            HIP_TRY(hipMemcpy(hp.data(), d->pos[d->cur], hp.size() * 8,
                              hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(hd.data(), d->drift[d->cur], hd.size() * 8,
                              hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(hl.data(), d->label[d->cur], hl.size() * 2,
                              hipMemcpyDeviceToHost));
            for (size_t s = 0; s < (size_t)nw; ++s)
                for (size_t i = 0; i < n; ++i) {
                    confs[(s * 2 + 0) * n + hl[s * n + i]] = hp[s * n + i];
                    confs[(s * 2 + 1) * n + hl[s * n + i]] = hd[s * n + i];
                }
        }
    }
    if (energy) {
        memset(energy, 0, W * sizeof(double));
        // after a step eslot[s] holds the parent's energy (actual.energy)
        HIP_TRY(hipMemcpy(energy, d->stepped ? d->eslot : d->energy[d->cur],
                          (size_t)nw * 8, hipMemcpyDeviceToHost));
    }
    if (weight) {
        for (size_t s = 0; s < W; ++s) weight[s] = s < (size_t)nw ? 1.0 : 0.0;
        if (!d->stepped)
            HIP_TRY(hipMemcpy(weight, d->weight[d->cur], (size_t)nw * 8,
                              hipMemcpyDeviceToHost));
    }
    if (mask)
        for (size_t s = 0; s < W; ++s) mask[s] = s < (size_t)nw ? 0 : 1;
    if (cloning_ref)
        for (size_t s = 0; s < W; ++s) cloning_ref[s] = (int64_t)href[s];
    if (scalars) {
        scalars[0] = c.e_t; scalars[1] = c.w_t; scalars[2] = c.ref_energy;
        scalars[3] = c.total_weight != 0.0 ? c.total_energy / c.total_weight
                                           : 0.0;
        scalars[4] = (double)nw;
    }
    return 0;
}

extern "C" int qmc_dmc_export_walkers(qmc_dmc *d, int64_t first, int64_t count,
                                      double *buf_dev)
{
    if (!d || !buf_dev) return fail("qmc_dmc_export_walkers: null argument");
    if (count <= 0) return 0;
    qmc_engine *e = d->eng;
    HIP_TRY(hipSetDevice(e->device));
    const int n = e->dm.n;
    long long tot = count * (long long)(3 * n + 2);
    hipLaunchKernelGGL(pack_walkers_kernel, dim3((unsigned)((tot + 255) / 256)),
                       dim3(256), 0, e->stream, d->pos[d->cur],
                       d->drift[d->cur], d->label[d->cur], d->energy[d->cur],
                       d->weight[d->cur],
                       (long long)first, (long long)count, n, buf_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int dmc_set_prev_nw(qmc_dmc *d, long long nw)
{
    // host-side patch of the control block (rebalance is a synchronising
    // operation anyway)
    DmcCtl c;
    HIP_TRY(hipMemcpyAsync(&c, d->ctl, sizeof(c), hipMemcpyDeviceToHost,
                           d->eng->stream));
    HIP_TRY(hipStreamSynchronize(d->eng->stream));
    c.prev_nw = nw;
    HIP_TRY(hipMemcpyAsync(d->ctl, &c, sizeof(c), hipMemcpyHostToDevice,
                           d->eng->stream));
    HIP_TRY(hipStreamSynchronize(d->eng->stream));
    return 0;
}

extern "C" int qmc_dmc_import_walkers(qmc_dmc *d, int64_t count,
                                      const double *buf_dev)
{
    if (!d || !buf_dev) return fail("qmc_dmc_import_walkers: null argument");
    if (count <= 0) return 0;
    qmc_engine *e = d->eng;
    HIP_TRY(hipSetDevice(e->device));
    int64_t nw = 0;
    int rc = qmc_dmc_num_walkers(d, &nw);
    if (rc) return rc;
    if (nw + count > d->maxw)
        return fail("qmc_dmc_import_walkers: population would exceed "
                    "max_num_walkers");
    const int n = e->dm.n;
    long long tot = count * (long long)(3 * n + 2);
    hipLaunchKernelGGL(unpack_walkers_kernel,
                       dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                       e->stream, d->pos[d->cur], d->drift[d->cur],
                       d->label[d->cur], d->energy[d->cur], d->weight[d->cur],
                       (long long)nw,
                       (long long)count, n, buf_dev);
    HIP_TRY(hipGetLastError());
    return dmc_set_prev_nw(d, nw + count);
}

extern "C" int qmc_dmc_truncate(qmc_dmc *d, int64_t new_nw)
{
    if (!d) return fail("qmc_dmc_truncate: null argument");
    HIP_TRY(hipSetDevice(d->eng->device));
    int64_t nw = 0;
    int rc = qmc_dmc_num_walkers(d, &nw);
    if (rc) return rc;
    if (new_nw < 0 || new_nw > nw)
        return fail("qmc_dmc_truncate: bad population size");
    return dmc_set_prev_nw(d, new_nw);
}
