// qmc_kernels_misc.h -- the kernels that do not depend on the lane-group shape:
// DMC branching (clone counts, scan, cloning table), E_ref feedback, the S(k) /
// density estimators, state gathers and the walker records of the population
// rebalance.  Included by qmcwalk.hip only (one definition per library).
#pragma once

#include "qmc_kernels.h"

struct BranchArgs {
    const double *weight;     // LOGARITHMS of the parent weights [maxw] (the
                              // device keeps log-weights: dmc_evolve_kernel)
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
            // a runaway weight must not overflow the conversion: no parent
            // can have more children than the population cap
            const double wc = fmin(exp(a.weight[s]) + u, (double)a.maxw);
            c = (int)wc;
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

// E_ref feedback (qmc_base/dmc.py:759-785) + per-step series; one thread.
__device__ __forceinline__ void dmc_finish_body(const FinishArgs &a, double e_sum)
{
    DmcCtl *c = a.ctl;
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

// Small populations (at most BR_FUSED_TILES tiles of 1024 parents, i.e. the
// reference's default 480 / 512 walkers): the whole branching step -- counts,
// scan, cloning table, E_t and W_t -- in ONE workgroup.  There a time step
// costs the device-side latency of its dependent launches (about 3 us each),
// not their work: 6 launches -> 3, 19.8 -> 15.0 us per step at 480 walkers.
// (Beyond two tiles the serial walk over the tiles loses: 4096 walkers
// 22 -> 32 us.)
static constexpr int BR_FUSED_TILES = 2;

// (`fin`: the bookkeeping of the PREVIOUS time step, which otherwise is a launch
// of its own between that step's drift-diffusion and this branching: inside a
// block of steps it rides at the head of this kernel -- 3 -> 2 dependent
// launches per step.)
__global__ void __launch_bounds__(BLOCK)
branch_fused_kernel(BranchArgs a, double *partial, FinishArgs fin, int do_fin)
{
    if (do_fin) {
        if (threadIdx.x == 0) dmc_finish_body(fin, 0.0);
        __threadfence_block();
        __syncthreads();
    }
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

__global__ void __launch_bounds__(BLOCK) dmc_finish_kernel(FinishArgs a)
{
    __shared__ double sh[BLOCK];
    double e_sum = 0.0;
    if (a.block_esum) e_sum = sum_block_esum(a.block_esum, a.ctl, sh);
    if (threadIdx.x != 0) return;
    dmc_finish_body(a, e_sum);
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
// doubles), energy, weight, then the walker's forward-walking estimator rows
// when estimators are enabled (S(k) parts [M][3], density [B]).
struct WalkerRecArgs {
    double *pos, *drift;
    unsigned short *label;
    double *energy, *weight;
    double *eslot;                  // slot energies
    double *ssf_aux, *dens_aux;     // current aux buffers or null
    long long first, count;
    int n, m3, nb;                  // particles, 3 * modes, bins
};

__device__ __forceinline__ int walker_rec_size(const WalkerRecArgs &a)
{
    return 3 * a.n + 2 + a.m3 + a.nb;
}

__global__ void pack_walkers_kernel(WalkerRecArgs a, double *buf)
{
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int rec = walker_rec_size(a), n = a.n;
    if (idx >= a.count * rec) return;
    long long s = idx / rec;
    int j = (int)(idx % rec);
    long long src = a.first + s;
    double v;
    if (j < n) v = a.pos[src * n + j];
    else if (j < 2 * n) v = a.drift[src * n + (j - n)];
    else if (j < 3 * n) v = (double)a.label[src * n + (j - 2 * n)];
    else if (j == 3 * n) v = a.energy[src];
    else if (j == 3 * n + 1) v = a.weight[src];
    else if (j < 3 * n + 2 + a.m3) v = a.ssf_aux[src * a.m3 + (j - 3 * n - 2)];
    else v = a.dens_aux[src * a.nb + (j - 3 * n - 2 - a.m3)];
    buf[idx] = v;
}

__global__ void unpack_walkers_kernel(WalkerRecArgs a, const double *buf)
{
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int rec = walker_rec_size(a), n = a.n;
    if (idx >= a.count * rec) return;
    long long s = idx / rec;
    int j = (int)(idx % rec);
    long long dst = a.first + s;
    double v = buf[idx];
    if (j < n) a.pos[dst * n + j] = v;
    else if (j < 2 * n) a.drift[dst * n + (j - n)] = v;
    else if (j < 3 * n) a.label[dst * n + (j - 2 * n)] = (unsigned short)v;
    else if (j == 3 * n) {
        a.energy[dst] = v;
        // the slot's previous occupant is gone: its "stale" energy (the
        // reference's quirk D1 reads it) becomes the newcomer's own
        a.eslot[dst] = v;
    }
    else if (j == 3 * n + 1) a.weight[dst] = v;
    else if (j < 3 * n + 2 + a.m3) a.ssf_aux[dst * a.m3 + (j - 3 * n - 2)] = v;
    else a.dens_aux[dst * a.nb + (j - 3 * n - 2 - a.m3)] = v;
}

// Population size after a rebalance (stream-ordered, no host round trip).
// Slots at or beyond `spare_cap` lose their cached Box-Muller normal: an
// imported walker must not consume the normal another walker stored there
// (the next odd step regenerates it from the slot's Philox block).
__global__ void dmc_set_nw_kernel(DmcCtl *ctl, long long nw,
                                  long long spare_cap)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ctl->prev_nw = nw;
        if (ctl->spare_nw > spare_cap) ctl->spare_nw = spare_cap;
    }
}

// label[s][i] = i: particles in their original order.
__global__ void ident_labels_kernel(unsigned short *label, long long rows, int n)
{
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < rows * n) label[idx] = (unsigned short)(idx % n);
}

// Replicate rows cyclically: dst[r] = src[r % src_rows] (initial populations
// larger than the VMC ensemble they are drawn from).
template <typename T>
__global__ void tile_rows_kernel(const T *src, long long src_rows, T *dst,
                                 long long dst_rows, int n)
{
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= dst_rows * n) return;
    long long r = idx / n;
    int i = (int)(idx % n);
    dst[idx] = src[(r % src_rows) * n + i];
}
