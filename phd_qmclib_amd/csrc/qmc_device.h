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

// Section markers.  Production build: nothing.  -DQMC_SECTIONS (tools/isa_one.sh):
// a comment in the ISA for the static instruction census.  -DQMC_TIMING (the
// diagnostic library of tools/section_times.py, never the shipped one): every
// mark reads the shader clock (s_memtime) and lane 0 adds the time since the
// wavefront's previous mark to the section that ends there -- the share of a
// wavefront's lifetime each section takes, measured while the kernel runs
// (qmc_engine_section_profile).
// -DQMC_CUTS (tools/section_counts.py, diagnostic): a wavefront ENDS at the
// mark whose id the host selected (qmc_engine_section_cut); instruction counters
// of runs cut at successive marks differ by what a section executes.
#define QMC_NSEC 32
#define QMC_SEC_COPIES 1024
#define QMC_NDIAG 4          // diagnostic counters (DevModel::diag)
#if defined(QMC_TIMING)
#define QMC_SECTION(name) qmc_stamp(m, qmc_sec_id(name))
#define QMC_SECTION_PHASE(off) qmc_stamp_phase(off)
#elif defined(QMC_CUTS)
#define QMC_SECTION(name)                                                     \
    do {                                                                      \
        if (m.sec_cut == qmc_sec_id(name) + QMC_SEC_OFF)                      \
            __builtin_amdgcn_endpgm();                                        \
    } while (0)
#define QMC_SECTION_PHASE(off) do { } while (0)
#elif defined(QMC_SECTIONS)
#define QMC_SECTION(name) asm volatile("; SECTION " name)
#define QMC_SECTION_PHASE(off) do { } while (0)
#else
#define QMC_SECTION(name) do { } while (0)
#define QMC_SECTION_PHASE(off) do { } while (0)
#endif

// ids of the sections (a second pass of eval_walker -- the energy pass of the
// VMC step -- is booked QMC_NSEC / 2 higher, qmc_stamp_phase)
constexpr const char *QMC_SEC_NAMES[QMC_NSEC / 2] = {
    "top", "load+philox+wrap", "resort", "tables+onebody", "pairs_in_lane",
    "leading_neighbour_steps", "leading_short_steps", "rotation",
    "rotation_loop_body", "rotation_last_step", "energy+logwf",
    "metropolis+store", "energy_pass", "store", "weight+store", "end" };

constexpr bool qmc_streq(const char *a, const char *b)
{
    return *a == *b && (*a == 0 || qmc_streq(a + 1, b + 1));
}
constexpr int qmc_sec_id(const char *name)
{
    for (int i = 0; i < QMC_NSEC / 2; ++i)
        if (qmc_streq(name, QMC_SEC_NAMES[i])) return i;
    return QMC_NSEC / 2 - 1;
}

struct DevModel {
    int n;                 // boson_number
    int ge;                // lanes of a group in use (padded shapes, see lane_particle)
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
    double am_cphi, am_sphi;   // |a_m| times them (qmc_sorted64.h)
    double cth, sth;       // cos(k2 L), |sin(k2 L)|
    int sth_sign;          // sign bit of sin(k2 L) (0 or 0x80000000)
    double sth_signed;     // sin(k2 L)
    // leading all-short rotation steps (eval_walker, one particle per lane): a
    // pair seen through the shifted tables is short, unwrapped and ordered iff
    //   sp_xlo < X < sp_xhi and Y > 0,   X = -k2 sin(theta), Y = cos(theta),
    // theta = k2 D' - phi; that identifies D' uniquely when k2 L < pi (sp_ok)
    double sp_xlo, sp_xhi;
    double sp_cos;         // cos(k2 rm)
    int sp_ok;
    // short-range variants (pair_core4): angles c_v added to k2 z_own,
    // v = (no wrap, d > 0), (no wrap, d < 0), (D > L/2), (D < -L/2)
    double var_cos[4], var_sin[4];
    double m_k2;           // -k2
    double m_k2_over_a;    // -k2 / a_long, a_long^2 (qmc_sorted64.h: Own64)
    double a_long_sq;
    double m_a_over_k2;    // -a_long / k2 (the tangent form, qmc_sorted64.h)
    double sin_rm;         // sin(pi rm / L)
    double a_long, b_long; // (pi/L) beta, (pi/L)^2 beta
    double inv_beta;       // 1 / beta
    double beta, log_am;
    double one_minus_beta; // 1 - beta (eval_sorted64: the merged logarithm)
    // one-body (Kronig-Penney)
    double z_a, z_b, k1, kp1, e0, v0, v0d, v0_minus_e0, cf;
    int uniform_barrier;   // every barrier has the same height v_barrier
    double v_barrier;
    double k1_2pi;         // k1 * 2 / pi
    double k1_half;        // k1 / 2
    // one-body table (one_body_tab): rows of OB_ROW doubles in device memory,
    // ob_m1 rows over the well [0, z_a], ob_m2 over the barrier [z_a, 1), one
    // closing row; null when the direct evaluation is used
    const double *ob_table;
    int ob_m1, ob_m2;
    double ob_invh1;       // m1 / z_a
    double ob_invh2;       // m2 / z_b   (>= ob_invh1)
    double ob_shift2;      // m1 / ob_invh2 - z_a
    // pair-table angles by row + angle addition (trig_tab): tg_rows rows of
    // width tg_h = L / tg_rows over [0, L), each {sin, cos(pi z_r / L), sin,
    // cos(k2 z_r)} at the row centre z_r; null when the model's angles span too
    // much for a cache-resident table
    const double *trig_table;
    int tg_rows;
    double tg_inv_h, tg_h;
    double tg_a1, tg_b1;   // pi/L, -(pi/L) h/2:  angle offset = a dz + b
    double tg_a2, tg_b2;   // k2,   -k2 h/2
    int sec_cut;           // QMC_CUTS builds: id of the mark where a wave ends
    // QMC_TIMING builds: QMC_SEC_COPIES x ([QMC_NSEC] cycles + [QMC_NSEC]
    // visits) (else null); a workgroup adds into copy blockIdx.x mod
    // QMC_SEC_COPIES (every wavefront adding into ONE set of counters
    // serialises on the atomics: the kernel ran 30x slower)
    unsigned long long *sec_prof;
    // diagnostic counters (qmc_engine_diag_counters; always allocated):
    // [0] walkers of the stepping kernels' sorted-row shapes that failed the
    // once-per-walker checks and were evaluated by eval_walker (the cold path;
    // counted there, nothing is added to the sorted-row path)
    unsigned long long *diag;
};

#if defined(QMC_TIMING)
struct QmcStampState {
    unsigned long long last[4];
    int cur[4], phase[4];
};
__device__ __forceinline__ QmcStampState &qmc_stamp_state()
{
    __shared__ QmcStampState st;
    return st;
}
__device__ __forceinline__ void qmc_stamp_phase(int off)
{
    if ((threadIdx.x & 63) == 0) qmc_stamp_state().phase[threadIdx.x >> 6] = off;
}
__device__ __forceinline__ void qmc_stamp(const DevModel &m, int id)
{
    const unsigned long long t = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0 && m.sec_prof) {
        QmcStampState &st = qmc_stamp_state();
        const int w = threadIdx.x >> 6;
        if (id != 0) {
            // (masked: a kernel without a "top" mark has no valid state)
            const int c = (st.cur[w] & (QMC_NSEC - 1)) +
                          2 * QMC_NSEC * (blockIdx.x & (QMC_SEC_COPIES - 1));
            atomicAdd(&m.sec_prof[c], t - st.last[w]);
            atomicAdd(&m.sec_prof[QMC_NSEC + c], 1ull);
        } else {
            st.phase[w] = 0;
        }
        st.cur[w] = id + st.phase[w];
        st.last[w] = t;
    }
}
#endif

// One-body table row (128 bytes, one cache line per particle): degree OB_DEG
// polynomials in t in [0, 1) across one interval of the unit cell,
// [0, 8) f1'/f1 and [8, 16) log f1.
#define OB_DEG 7
#define OB_ROW 16

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
    // ((hi >> 5) 2^26 + (lo >> 6)) 2^-53, assembled from two exact 32-bit
    // conversions (6 instructions; through a 64-bit integer the compiler
    // needs 12) -- every step is exact, the value is the same
    const double a = (double)(hi >> 5);     // 27 bits
    const double b = (double)(lo >> 6);     // 26 bits
    return fma(a, 0x1p-27, b * 0x1p-53);
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

// Philox2x32-10 (same paper): one 64-bit block per call.  The uniform proposal
// of the VMC step draws from it: a particle needs ONE number per step and a
// wavefront pays per instruction issued, not per lane served -- the 4x32
// generator computed 128 bits per lane to use 64 of them (18 wide multiplies
// and 34 xors per chain-step against 10 and 20 here).
__device__ __forceinline__ void philox2x32_10(uint32_t &c0, uint32_t &c1,
                                              uint32_t k)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p = (uint64_t)0xD256D193u * c0;
        const uint32_t n0 = (uint32_t)(p >> 32) ^ k ^ c1;
        c1 = (uint32_t)p;
        c0 = n0;
        k += 0x9E3779B9u;
    }
}

// The VMC move stream (same definition in oracle/qmc_oracle.c:
// orc_vmc_move_block): block of particle `index` of chain `slot` at Metropolis
// step `step`.  Every (slot < 2^28, step < 2^26, index < 1024) has its own
// counter under the key of the seed; bits beyond those ranges move into the
// key.  Word 0 moves the particle -- (w0 + 1/2) 2^-32 - 1/2 times the move
// spread, exact in double -- and the second words of the blocks of particles 0
// and 1 make the step's accept draw (53 bits).
__device__ __forceinline__ void vmc_move_block(uint64_t seed, uint32_t slot,
                                               uint32_t step, uint32_t index,
                                               uint32_t &w0, uint32_t &w1)
{
    uint32_t key = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x85EBCA6Bu);
    key += (step >> 26) * 0x632BE5ABu + (slot >> 28) * 0xC2B2AE35u;
    w0 = ((step & 0x3FFFFFFu) << 6) | ((slot >> 22) & 0x3Fu);
    w1 = ((slot & 0x3FFFFFu) << 10) | (index & 0x3FFu);
    philox2x32_10(w0, w1, key);
}

__device__ __forceinline__ double vmc_move_unit(uint32_t w0)
{
    // (one conversion and one fused multiply-add: 2^-33 - 1/2 is exact)
    return fma((double)w0, 0x1p-32, 0x1p-33 - 0.5);
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

// The DMC diffusion stream (same definition in oracle/qmc_oracle.c:
// orc_dmc_normal): one Philox2x32-10 block per (walker slot, pair of time
// steps, particle) -- the counter packing of the VMC move blocks under another
// key -- and both Box-Muller normals of it: the cosine branch moves the
// particle at time step 2m, the sine branch at 2m + 1.  The two uniforms carry
// 32 bits each, (w + 1/2) 2^-32 in (0, 1): a normal is cut off at 6.76 sigma
// (2^-33 of the radial weight) and its angle is resolved to 1.5e-9.  Round 4
// drew 2 x 53 bits from a Philox4x32 block here: 20 wide multiplies and 40
// xors per block instead of 10 and 20, four conversions instead of two -- and
// two particles per lane (no spare row) draw a block per particle and step.
__device__ __forceinline__ void dmc_normal2(uint64_t seed, uint32_t slot,
                                            uint32_t step2, uint32_t index,
                                            double &g0, double &g1)
{
    uint32_t key = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x85EBCA6Bu);
    key += 0x27D4EB2Fu;      // (apart from the VMC move blocks of the seed)
    key += (step2 >> 26) * 0x632BE5ABu + (slot >> 28) * 0xC2B2AE35u;
    uint32_t w0 = ((step2 & 0x3FFFFFFu) << 6) | ((slot >> 22) & 0x3Fu);
    uint32_t w1 = ((slot & 0x3FFFFFu) << 10) | (index & 0x3FFu);
    philox2x32_10(w0, w1, key);
    // (one conversion and one fused multiply-add each, exact)
    const double u0 = fma((double)w0, 0x1p-32, 0x1p-33);
    const double u1 = fma((double)w1, 0x1p-32, 0x1p-33);
    // u0 <= 1 - 2^-33: the logarithm is < 0
    const double r = fast_sqrt(-2.0 * log_pos(u0));
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

// Slot (index into the walker's row of positions) held by (lane gl, register p)
// of a lane group; >= n = none.
// One walker per wavefront with several particles per lane (N > 64): a lane
// holds P CONSECUTIVE slots, i = P gl + p.  The rows are kept in ascending
// position, so the partner lane of rotation step k holds the particles about
// P k places away -- all P^2 pairs of a lane at a step are at about the same
// separation and the short-range branch is taken or skipped by whole
// wavefronts.  (With slots strided over the lanes, i = gl + 64 p, every step
// mixes near and far partners: N = 128 ran both branches in all 31 steps.)
// Smaller groups: i = gl + G p.  A padded shape (N < G P) uses only the first
// ge = 2 ceil(N / 2P) lanes of the group (strided: i = gl + ge p) and rotates
// over those: N = 100 on the (64, 2) shape runs 25 rotation steps over 50 lanes
// instead of 32 over 64 (round 1 paid the full shape: 4.7e11 pair evaluations/s
// at N = 100 against 8.0e11 at N = 128).
#ifndef QMC_INTERLEAVE
#define QMC_INTERLEAVE 1
#endif
template <int G, int P>
struct SlotMap {
    static constexpr bool CONSECUTIVE = QMC_INTERLEAVE && (G == 64) && (P >= 2);
};

template <int G, bool PAD>
__device__ __forceinline__ int lanes_in_use(const DevModel &m)
{
    return PAD ? m.ge : G;
}

template <int G, int P, bool PAD>
__device__ __forceinline__ int lane_particle(const DevModel &m, int gl, int p)
{
    if (SlotMap<G, P>::CONSECUTIVE) return P * gl + p;
    if (!PAD) return gl + G * p;
    const int ge = m.ge;
    return gl < ge ? gl + ge * p : m.n;
}

// Sum over the lanes of a group, in every lane.  Inside a row of 16 lanes four
// DPP rotations (row_ror 8, 4, 2, 1: two v_mov_dpp + one add each, no LDS);
// across rows the xor butterfly through ds_bpermute.
template <int N>
__device__ __forceinline__ double row_ror_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 + N,
                                               0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 + N,
                                               0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <int G>
__device__ __forceinline__ double group_sum(double v)
{
    static_assert(G >= 16, "lane groups are 16, 32 or 64 lanes");
    v += row_ror_f64<8>(v);
    v += row_ror_f64<4>(v);
    v += row_ror_f64<2>(v);
    v += row_ror_f64<1>(v);
#pragma unroll
    for (int m = 16; m < G; m <<= 1)
        v += __shfl_xor(v, m, 64);
    return v;
}

// The same over a whole wavefront on the matrix cores: a v_mfma_f64_16x16x4
// contracts over k = lane >> 4, so A = the lanes' values against B = ones
// leaves r_i = the sum of the four lanes with lane & 15 = i in row i; a lane
// holds rows (lane >> 4) + 4 reg, adds its four and feeds the partial back as
// A: every element of the second product is the total.  Two sums share the
// second product (rows 0-7 carry one, rows 8-15 the other).  Against the
// butterfly (12 ds_bpermute + 6 adds per sum, the LDS pipe being the kernel's
// second-busiest resource) this is 3 vector adds and no LDS traffic.
#ifndef QMC_MFMA_SUM
#define QMC_MFMA_SUM 1
#endif
typedef double qmc_v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_partial_mfma(double v)
{
    const qmc_v4d zero = {0.0, 0.0, 0.0, 0.0};
    const qmc_v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(v, 1.0, zero, 0, 0, 0);
    return (d[0] + d[1]) + (d[2] + d[3]);
}

// One sum over the wavefront through the 4x4x4 form (four independent 4x4
// blocks; 16 cycles an instruction against 64 for 16x16x4, profiles/
// r02_ubench3_mfma_overlap.txt).  Lane layout, probed on the chip
// (tools/mfma4_layout.hip): block b = (lane / 4) % 4; A[i][k] sits in lane
// i + 4 b + 16 k and D[i][j] in lane j + 4 b + 16 i.  Against B = ones the first
// product leaves, in row i of the wavefront, the sum over the four rows of
// the lanes i + 4 b; fed back as A the second leaves the total of the block's
// 16 lanes {i + 4 b + 16 k} in all of them; two rotations inside the rows of
// 16 (by 4 and by 8 lanes: DPP, no LDS) add the four blocks.
#ifndef QMC_MFMA4_SUM
#define QMC_MFMA4_SUM 1
#endif
template <int N> __device__ __forceinline__ double row_ror_f64(double v);
__device__ __forceinline__ double wave_sum_mfma4(double v)
{
    const double r = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
    double t = __builtin_amdgcn_mfma_f64_4x4x4f64(r, 1.0, 0.0, 0, 0, 0);
    t += row_ror_f64<4>(t);
    t += row_ror_f64<8>(t);
    return t;
}

__device__ __forceinline__ double wave_sum_mfma(double v)
{
    if (QMC_MFMA4_SUM) return wave_sum_mfma4(v);
    const qmc_v4d zero = {0.0, 0.0, 0.0, 0.0};
    const qmc_v4d t = __builtin_amdgcn_mfma_f64_16x16x4f64(
        wave_partial_mfma(v), 1.0, zero, 0, 0, 0);
    return t[0];
}

__device__ __forceinline__ void wave_sum2_mfma(double a, double b, double &sa,
                                               double &sb)
{
    if (QMC_MFMA4_SUM) {
        sa = wave_sum_mfma4(a);
        sb = wave_sum_mfma4(b);
        return;
    }
    const qmc_v4d zero = {0.0, 0.0, 0.0, 0.0};
    const double pa = wave_partial_mfma(a), pb = wave_partial_mfma(b);
    const qmc_v4d t = __builtin_amdgcn_mfma_f64_16x16x4f64(
        (threadIdx.x & 8) ? pb : pa, 1.0, zero, 0, 0, 0);
    sa = t[0];          // rows 0-3
    sb = t[2];          // rows 8-11
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
                                            double L, double half_L,
                                            int ge = G)
{
    const int lane = threadIdx.x & 63, base = lane - gl;
    // partner in the row: even phase (0,1)(2,3)..; odd phase (1,2)(3,4)..
    // and, across the row seam, (ge-1 of row a, 0 of row a+1) cyclically
    // (ge, the lanes in use, is even: every lane has exactly one partner)
    int pg = (parity & 1u) ? ((gl & 1) ? gl + 1 : gl - 1) : (gl ^ 1);
    const bool wrap_hi = pg >= ge, wrap_lo = pg < 0;
    if (wrap_hi) pg = 0;
    if (wrap_lo) pg = ge - 1;
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
        const bool valid = gl < ge && (gl + ge * a) < n &&
                           (pg + ge * want) < n;
        // "lower" = the element whose rank comes first in the cyclic order
        const bool lower = wrap_hi ? true : (wrap_lo ? false : gl < pg);
        double d = lower ? zp - z0[a] : z0[a] - zp;   // upper minus lower
        if (d > half_L) d -= L;
        if (d < -half_L) d += L;
        if (valid && d < 0.0) { z[a] = zp; lab[a] = lp; }
    }
}

// One walker per wavefront, one particle per lane (N <= 64): the lanes hold the
// particles in ASCENDING position, not merely in cyclic order -- the place where
// the positions wrap from L back to 0 stays at the lane-index seam (lane ge-1
// -> lane 0).  The shifted second copy of the pair tables (pair_core1) relies
// on it for speed (never for correctness): with the wrap point anywhere else,
// 2k lanes of rotation step k leave the fast case.  Two pieces keep the order: an odd-even transposition pass on plain
// positions that never exchanges across the seam, and `anchor_seam`, which
// notices a particle that crossed the box boundary (it now sits at the wrong
// end of the lanes) and rotates the whole row by one lane.
template <int G>
__device__ __forceinline__ void resort_linear(double &z, int &lab, int gl,
                                              unsigned parity, int ge)
{
    const int lane = threadIdx.x & 63, base = lane - gl;
    const int pg = (parity & 1u) ? ((gl & 1) ? gl + 1 : gl - 1) : (gl ^ 1);
    const bool has = gl < ge && pg >= 0 && pg < ge;
    const int src = base + (has ? pg : gl);
    const double zp = __shfl(z, src, 64);
    const int lp = __shfl(lab, src, 64);
    // the lower lane of a pair keeps the smaller position
    const bool take = has && ((gl < pg) ? zp < z : zp > z);
    if (take) { z = zp; lab = lp; }
}

// Travelling sums of the rotation loop move one lane per step.  On a whole
// wavefront that is `v_mov_b32_dpp wave_ror:1` (two per double) instead of two
// ds_bpermute_b32 through the LDS pipe, which the loop's table reads need:
// -3 % on the VMC and DMC steps at N = 64 (profiles/r02_ab_variants.txt).
// (Adding the partner's share into an LDS row with ds_add_f64 instead -- no
// travelling sum at all, 3 vector instructions fewer per step -- measured the
// same time with two table-shaped rows and 3.5 % more with one masked row.)
#ifndef QMC_T_DPP
#define QMC_T_DPP 1
#endif
// every lane takes the value of the lane below it in its group, the first lane
// that of the last (groups of 64: wave_ror:1; of 16: row_ror:1)
template <int G>
__device__ __forceinline__ int group_ror1_b32(int x)
{
    static_assert(G == 64 || G == 16, "a DPP rotation exists for 16 and 64 lanes");
    if (G == 64) return __builtin_amdgcn_update_dpp(x, x, 0x13C, 0xf, 0xf, false);
    return __builtin_amdgcn_update_dpp(x, x, 0x121, 0xf, 0xf, false);
}
template <int G>
__device__ __forceinline__ double group_ror1(double v)
{
    return __hiloint2double(group_ror1_b32<G>(__double2hiint(v)),
                            group_ror1_b32<G>(__double2loint(v)));
}
template <int G>
__device__ __forceinline__ float group_ror1(float v)
{
    return __int_as_float(group_ror1_b32<G>(__float_as_int(v)));
}

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// (G = 64: the row is the wavefront, the test is scalar and the rotation a
// wave-uniform branch that is almost never taken.)
__device__ __forceinline__ void anchor_seam(double &z, int &lab,
                                            int ge /* lanes holding particles */)
{
    const int lane = threadIdx.x & 63;
    const double z0 = readlane_f64(z, 0), zl = readlane_f64(z, ge - 1);
    if (zl < z0) {
        const double z1 = readlane_f64(z, 1), zl1 = readlane_f64(z, ge - 2);
        int src = lane;
        if (zl < z1) {
            // the last lane's particle left through z = L and is now the
            // smallest: everybody moves one lane up, it takes lane 0
            if (lane < ge) src = lane == 0 ? ge - 1 : lane - 1;
        } else if (z0 > zl1) {
            // the first lane's particle left through z = 0: the other way
            if (lane < ge) src = lane == ge - 1 ? 0 : lane + 1;
        }
        z = __shfl(z, src, 64);
        lab = __shfl(lab, src, 64);
    }
}

// The same for several particles per lane (consecutive slots, SlotMap): the
// sequence is the slot index i = P gl + a.  Even phase: pairs (i, i + 1) with i
// even, both in one lane (P is even) -- no lane traffic; odd phase: the pairs
// inside the lane and the one across to the next lane.
template <int P>
__device__ __forceinline__ void order_pair(double &za, int &la, double &zb,
                                           int &lb, bool valid)
{
    const bool sw = valid && zb < za;
    const double t = za; const int u = la;
    za = sw ? zb : za; la = sw ? lb : la;
    zb = sw ? t : zb; lb = sw ? u : lb;
}

template <int P>
__device__ __forceinline__ void resort_linear_rows(double (&z)[P], int (&lab)[P],
                                                   int gl, unsigned parity, int n)
{
    static_assert(P % 2 == 0, "consecutive slots: an even number per lane");
    const int i0 = P * gl;
    if (!(parity & 1u)) {
#pragma unroll
        for (int a = 0; a + 1 < P; a += 2)
            order_pair<P>(z[a], lab[a], z[a + 1], lab[a + 1], i0 + a + 1 < n);
    } else {
        const int lane = threadIdx.x & 63;
        // the neighbours' end slots as they are now
        const double up = __shfl(z[0], lane + 1, 64);
        const int lup = __shfl(lab[0], lane + 1, 64);
        const double dn = __shfl(z[P - 1], lane - 1, 64);
        const int ldn = __shfl(lab[P - 1], lane - 1, 64);
        const double top = z[P - 1], bot = z[0];
#pragma unroll
        for (int a = 1; a + 1 < P; a += 2)
            order_pair<P>(z[a], lab[a], z[a + 1], lab[a + 1], i0 + a + 1 < n);
        // (slot i0 + P - 1, slot i0 + P) and (slot i0 - 1, slot i0): both
        // sides of a pair see the same two values
        if (gl < 63 && i0 + P < n && up < top) { z[P - 1] = up; lab[P - 1] = lup; }
        if (gl > 0 && i0 < n && dn > bot) { z[0] = dn; lab[0] = ldn; }
    }
}

// register r (wave-uniform, runtime) of a lane's row
template <int P, typename T>
__device__ __forceinline__ T row_reg(const T (&v)[P], int r)
{
    T x = v[0];
#pragma unroll
    for (int a = 1; a < P; ++a) x = (r == a) ? v[a] : x;
    return x;
}

template <int P>
__device__ __forceinline__ void anchor_seam_rows(double (&z)[P], int (&lab)[P],
                                                 int n)
{
    const int lane = threadIdx.x & 63;
    const int ln = (n - 1) / P, rn = (n - 1) % P;      // the last slot
    const double z0 = readlane_f64(z[0], 0);
    const double zl = readlane_f64(row_reg<P>(z, rn), ln);
    if (zl < z0) {
        const double z1 = readlane_f64(z[1], 0);
        const int ln2 = (n - 2) / P, rn2 = (n - 2) % P;
        const double zl1 = readlane_f64(row_reg<P>(z, rn2), ln2);
        if (zl < z1) {
            // the last slot's particle left through z = L and is now the
            // smallest: every particle moves one slot up, it takes slot 0
            const int labl = __builtin_amdgcn_readlane(row_reg<P>(lab, rn), ln);
            double zin = __shfl(z[P - 1], lane - 1, 64);
            int lin = __shfl(lab[P - 1], lane - 1, 64);
            if (lane == 0) { zin = zl; lin = labl; }
#pragma unroll
            for (int a = P - 1; a >= 1; --a) { z[a] = z[a - 1]; lab[a] = lab[a - 1]; }
            z[0] = zin; lab[0] = lin;
        } else if (z0 > zl1) {
            // the first slot's particle left through z = 0: the other way
            const int lab0 = __builtin_amdgcn_readlane(lab[0], 0);
            const double zin = __shfl(z[0], lane + 1, 64);
            const int lin = __shfl(lab[0], lane + 1, 64);
#pragma unroll
            for (int a = 0; a + 1 < P; ++a) { z[a] = z[a + 1]; lab[a] = lab[a + 1]; }
            z[P - 1] = zin; lab[P - 1] = lin;
#pragma unroll
            for (int a = 0; a < P; ++a)
                if (lane == ln && a == rn) { z[a] = z0; lab[a] = lab0; }
        }
    }
}


// Per-particle table entry kept in registers by the owner and published to LDS.
// R is the arithmetic type of the pair loop: double, or float for the
// reduced-precision variant (the reference's `jit_fastmath` knob,
// mrbp_qmc/dmc.py:159-160; never the default).  Positions, the per-particle
// tables' sin/cos, the one-body factor, the energy assembly and the logarithms
// stay in double either way.
template <typename R>
struct PTabT {
    R s, c;    // sin/cos(pi z / L)
    R su, cu;  // sin/cos(k2 z)
};
typedef PTabT<double> PTab;

// sin/cos of the two pair-table angles of a particle at z from the row table:
// theta = theta_row + delta with |delta| <= QMC_TRIG_DMAX (host: build_trig_table),
// sin/cos(delta) by their first terms (truncation < 1e-17), the row values
// correctly rounded.  29 vector instructions against 68 for two polynomial
// `sincos_halfpi`.  Returns false -- for the whole wavefront -- if any lane's z
// lies outside [0, L) (the batch evaluation takes positions as they come); the
// caller then evaluates directly.
typedef const __attribute__((address_space(1))) double *qmc_gptr;

__device__ __forceinline__ void sincos_small(double d, double &s, double &c)
{
    const double d2 = d * d;
    s = fma(d * d2, fma(d2, sconst(1.0 / 120.0), sconst(-1.0 / 6.0)), d);
    c = fma(d2, fma(d2, sconst(1.0 / 24.0), -0.5), 1.0);
}

struct TrigRow {
    double S1, C1, S2, C2;   // the row
    double dz;               // position inside it
};

// the row's loads, issued early (the one-body factor is evaluated under their
// latency); false -- for the whole wavefront -- if a lane is outside the table
__device__ __forceinline__ bool trig_tab_load(const DevModel &m, double z,
                                              TrigRow &t)
{
    const int r = (int)(z * m.tg_inv_h);          // truncation
    if (__ballot((unsigned)r >= (unsigned)m.tg_rows)) return false;
    t.dz = fma(-(double)r, m.tg_h, z);            // [0, h): r h is exact
    const qmc_gptr row = (qmc_gptr)m.trig_table + 4u * (unsigned)r;
    t.S1 = row[0]; t.C1 = row[1]; t.S2 = row[2]; t.C2 = row[3];
    return true;
}

__device__ __forceinline__ void trig_tab_finish(const DevModel &m,
                                                const TrigRow &t, PTab &ta)
{
    double sd, cd;
    sincos_small(fma(t.dz, m.tg_a1, m.tg_b1), sd, cd);
    ta.s = fma(t.S1, cd, t.C1 * sd);
    ta.c = fma(t.C1, cd, -(t.S1 * sd));
    sincos_small(fma(t.dz, m.tg_a2, m.tg_b2), sd, cd);
    ta.su = fma(t.S2, cd, t.C2 * sd);
    ta.cu = fma(t.C2, cd, -(t.S2 * sd));
}

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

// The same from the table.  The one-body factor is a fixed function of the
// position inside the unit cell: f1'/f1 and log f1 are tabulated per model as
// piecewise degree-7 polynomials (host: build_ob_table in qmcwalk.hip, which
// verifies them against the closed forms to ~1e-15 and falls back to the
// direct evaluation when a model does not reach that).  Both lattice regions
// cost the same ~25 instructions here; evaluated directly the two transcendental
// branches run one after the other in nearly every wave (~140).  The rows
// (<= 64 KB) are read through the vector cache; the loads are issued before
// pair tables occupy registers.
// `logf1` is log of the factor itself (the direct path returns the factor and
// a split-off exponent).
template <bool WF, bool LDZ = true>
__device__ __forceinline__ void one_body_tab(const DevModel &m, double z,
                                             double &ldz, double &logf1,
                                             bool &barrier)
{
    const double zc = __builtin_amdgcn_fract(z);
    // the region decides -f1''/f1 + V, which jumps at z_a: an exact compare,
    // as in the reference (mrbp_qmc/model.py:464, 549), not the row index
    barrier = m.z_a < zc;
    // row + position inside it.  The map is piecewise linear (slope m1 / z_a
    // over the well, m2 / z_b >= that over the barrier, continuous at z_a),
    // i.e. the larger of its two lines: no compare, no select.  (f1'/f1 and
    // log f1 are continuous at z_a, so a last-bit difference between this
    // and the compare above is harmless.)
    const double u = fmax(zc * m.ob_invh1, (zc + m.ob_shift2) * m.ob_invh2);
    // 32-bit row offset against the uniform base: global loads with a scalar
    // base address (a generic pointer would make them flat loads)
    const unsigned off = (unsigned)(int)u * OB_ROW;
    const double t = __builtin_amdgcn_fract(u);
    const qmc_gptr r = (qmc_gptr)m.ob_table + off;
    if (LDZ) {
        double p = r[OB_DEG];
#pragma unroll
        for (int k = OB_DEG - 1; k >= 0; --k) p = fma(p, t, r[k]);
        ldz = p;
    }
    if (WF) {
        double q = r[8 + OB_DEG];
#pragma unroll
        for (int k = OB_DEG - 1; k >= 0; --k) q = fma(q, t, r[8 + k]);
        logf1 = q;
    }
}

// -f1''/f1 + V(z) of a particle (added to (f1'/f1)^2): e0 in the well,
// V - (V0 - e0) in a barrier of height V (mrbp_qmc/model.py:446-464, 533-551).
__device__ __forceinline__ double one_body_kin_const(const DevModel &m,
                                                     double z, bool barrier)
{
    double v = m.v_barrier;
    if (!m.uniform_barrier) {
        // lattice defects: every defects_sep-th barrier has height v0d
        // (floor and a non-negative remainder: callers of the batch
        // evaluation may pass positions outside [0, L))
        int rr = (int)floor(z) % m.defects_sep;
        if (rr < 0) rr += m.defects_sep;
        v = (rr == 0) ? m.v0d : m.v0;
    }
    return barrier ? v - m.v0_minus_e0 : m.e0;
}

// Constants of the pair loop, loaded once per walker evaluation.  The two that
// feed a copysign live in VGPRs (a v_bfi on the high dword then needs no move
// of the low dword from an SGPR every pair).
template <typename R>
struct PairConstsT {
    R sin_rm, cth, m_k2cphi, sphi, cphi;
    R v_sth, v_k2sphi;       // VGPR-resident
    R half_L, rm, L_minus_rm;
    R m_k2;
    int sth_sign;
    R sth_signed, k2sphi;
    R k2sq, inv_beta, b_long;
    R sp_xlo, sp_xhi, sp_cos;
    int sp_ok;
};
typedef PairConstsT<double> PairConsts;

template <typename R>
__device__ __forceinline__ PairConstsT<R> load_pair_consts(const DevModel &m)
{
    PairConstsT<R> c;
    c.sin_rm = (R)m.sin_rm; c.cth = (R)m.cth; c.m_k2cphi = (R)m.m_k2cphi;
    c.sphi = (R)m.sphi; c.cphi = (R)m.cphi;
    c.half_L = (R)m.half_L; c.rm = (R)m.rm; c.L_minus_rm = (R)m.L_minus_rm;
    c.v_sth = (R)m.sth; c.v_k2sphi = (R)m.k2sphi;
    c.sth_sign = m.sth_sign;
    c.m_k2 = (R)m.m_k2;
    c.sth_signed = (R)m.sth_signed; c.k2sphi = (R)m.k2sphi;
    c.k2sq = (R)m.k2sq; c.inv_beta = (R)m.inv_beta; c.b_long = (R)m.b_long;
    c.sp_xlo = (R)m.sp_xlo; c.sp_xhi = (R)m.sp_xhi; c.sp_ok = m.sp_ok;
    c.sp_cos = (R)m.sp_cos;
    asm volatile("" : "+v"(c.v_sth), "+v"(c.v_k2sphi));
    return c;
}

// Short-range X, Y of a pair from the k2-tables of both particles, exact for
// any order of the two and either side of the periodic wrap.
template <typename R, bool EN = true>
__device__ __forceinline__ void short_generic(const PairConstsT<R> &m,
                                              const PTabT<R> &a,
                                              const PTabT<R> &b, R S,
                                              bool wrapped, R &X, R &Y)
{
    R Su = a.su * b.cu - a.cu * b.su;   // sin(k2 (z_a - z_b))
    R Cu = a.cu * b.cu + a.su * b.su;
    if (wrapped) {
        // keep this a real (exec-masked) branch: as selects it costs four
        // v_cndmask on top of the arithmetic
        asm volatile("");
        // min image d = D - sgn(D) L; sgn(D) = sgn(S)
        if constexpr (sizeof(R) == 8) {
            // t = sin(k2 L) sgn(S): copysign on |sin(k2 L)|, then the sign
            // of sin(k2 L) itself (k2 L is any angle) xor-ed into the high
            // word
            double t = __builtin_copysign(m.v_sth, S);
            t = __hiloint2double(__double2hiint(t) ^ m.sth_sign,
                                 __double2loint(t));
            double ct, st;
            const double cth = m.cth;
            // in place, exactly four instructions (the compiler's
            // two-address v_fmac form needs two extra 64-bit moves at the
            // join)
            asm("v_mul_f64 %[ct], %[cu], %[t]\n\t"
                "v_mul_f64 %[st], %[su], %[t]\n\t"
                "v_fma_f64 %[su], %[su], %[cth], -%[ct]\n\t"
                "v_fma_f64 %[cu], %[cu], %[cth], %[st]"
                : [su] "+v"(Su), [cu] "+v"(Cu), [ct] "=&v"(ct),
                  [st] "=&v"(st)
                : [t] "v"(t), [cth] "s"(cth));
        } else {
            const R t = q_copysign(m.sth_signed, S * m.sth_signed);
            const R ns = Su * m.cth - Cu * t;
            const R nc = Cu * m.cth + Su * t;
            Su = ns; Cu = nc;
        }
    }
    // now (Su, Cu) = sin/cos(k2 d), |k2 d| < pi/2, sgn(Su) = sgn(d):
    //   -k2 tan(k2 r - phi) sgn(d) = X / Y with
    if (EN) {
        R t2 = q_copysign(m.v_k2sphi, Su);
        X = q_fma(m.m_k2cphi, Su, Cu * t2);
    }
    Y = q_fma(q_abs(Su), m.sphi, Cu * m.cphi);
}

// One pair, seen from the own particle (table `a`, long-range numerator
// coefficients aks/akc = a_long * (sin, cos)) against partner table `b`.
//   q       : contribution to the drift of the own particle (partner: -q)
//   Yout    : the denominator = the factor |f2| up to constants
//             (short: cos(k2 r - phi); long: sin(pi d / L), signed)
//   isshort : r < rm
//   live    : both particles exist.  The idle slots of a padded shape hold
//             z = 0; classified like real particles they are "short" whenever
//             their partner is near the box boundary and pull whole wavefronts
//             through the short-range branches (N = 100: 15 % of the step).
//   EN      : the quotient is wanted (energy / drift); without it only the
//             factor Yout and the class (the log|psi|-only pass of the VMC step)
template <bool ZCLASS, typename R, bool EN = true>
__device__ __forceinline__ void pair_core(const PairConstsT<R> &m,
                                          const PTabT<R> &a, R aks, R akc,
                                          R za, const PTabT<R> &b, R zb,
                                          bool live, R &q, R &Yout,
                                          bool &isshort,
                                          unsigned long long &shortmask)
{
    R S = a.s * b.c - a.c * b.s;     // sin(pi (z_a - z_b) / L)
    R X = 0;
    if (EN) X = akc * b.c + aks * b.s;   // a_long * cos(...)
    R Y = S;
    bool wrapped = false;
    if (ZCLASS) {
        R aD = q_abs(za - zb);
        wrapped = aD > m.half_L;
        isshort = live & ((aD < m.rm) | (aD > m.L_minus_rm));
    } else {
        if (EN) wrapped = X < (R)0;       // |z_a - z_b| > L/2
        isshort = live & (q_abs(S) < m.sin_rm);   // min-image r < rm
    }
    // taken here, in the block of the compare, the ballot is the compare's own
    // SGPR mask (later it costs a v_cndmask + v_cmp round trip)
    shortmask = __ballot(isshort);
    if (isshort) {
        // (without the quotient the cosine is formed for the short pairs only)
        if (!EN && !ZCLASS) wrapped = (a.c * b.c + a.s * b.s) < (R)0;
        short_generic<R, EN>(m, a, b, S, wrapped, X, Y);
    }
    if (EN) q = pair_div(X, Y);
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
template <typename R>
struct ShortTabT {
    R s[4], c[4];
};
typedef ShortTabT<double> ShortTab;

template <typename R>
__device__ __forceinline__ void make_short_tab(const DevModel &m, double su,
                                               double cu, ShortTabT<R> &st)
{
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        st.s[v] = (R)fma(su, m.var_cos[v], cu * m.var_sin[v]);
        st.c[v] = (R)fma(cu, m.var_cos[v], -(su * m.var_sin[v]));
    }
}

template <bool ZCLASS, typename R, bool EN = true>
__device__ __forceinline__ void pair_core4(const PairConstsT<R> &m,
                                           const PTabT<R> &a,
                                           const ShortTabT<R> &sa, R aks,
                                           R akc, R za, const PTabT<R> &b,
                                           R zb, bool live, R &q, R &Yout,
                                           bool &isshort,
                                           unsigned long long &shortmask)
{
    R S = a.s * b.c - a.c * b.s;     // sin(pi (z_a - z_b) / L)
    R X = 0;
    if (EN) X = akc * b.c + aks * b.s;   // a_long * cos(...)
    R Y = S;
    bool wrapped = false, neg;
    if (ZCLASS) {
        const R D = za - zb;
        const R aD = q_abs(D);
        wrapped = aD > m.half_L;
        neg = D < (R)0;
        isshort = live & ((aD < m.rm) | (aD > m.L_minus_rm));
    } else {
        if (EN) wrapped = X < (R)0;       // |z_a - z_b| > L/2
        neg = S < (R)0;                   // sgn(D) = sgn(sin(pi D / L))
        isshort = live & (q_abs(S) < m.sin_rm);   // min-image r < rm
    }
    shortmask = __ballot(isshort);
    if (isshort) {
        R xs = 0, ys;
        if (!EN && !ZCLASS) wrapped = (a.c * b.c + a.s * b.s) < (R)0;
        // real (exec-masked) branches: as selects the four cases would cost
        // eight v_cndmask per double pair
#define QMC_CASE4(v)                                                          \
        {                                                                     \
            asm volatile("");                                                 \
            if (EN) xs = sa.s[v] * b.cu - sa.c[v] * b.su;                     \
            ys = sa.c[v] * b.cu + sa.s[v] * b.su;                             \
        }
        if (!wrapped) {
            if (!neg) QMC_CASE4(0) else QMC_CASE4(1)
        } else {
            if (!neg) QMC_CASE4(2) else QMC_CASE4(3)
        }
#undef QMC_CASE4
        if (EN) X = m.m_k2 * xs;
        Y = ys;
    }
    if (EN) q = pair_div(X, Y);
    Yout = Y;
}

// Short-range pair, two-case form (two particles per lane, rows in ascending
// order).  Of the four (wrap, sign) cases a sorted row meets two: the partner a
// few places below (D in (0, L/2)) and, in the lanes whose partner index wraps,
// the partner at the far end of the box (D in (-L, -L/2)).  The own particle
// carries the angle tables of those two (8 registers instead of 32 for four, 8
// instead of 16 instructions to build them); a pair in another case -- two
// neighbours that passed each other since the last sort pass -- takes
// `short_generic`.
template <typename R>
struct ShortTab2T {
    R s0, c0;      // (no wrap, d > 0)
    R s3, c3;      // (D < -L/2)
};

template <typename R>
__device__ __forceinline__ void make_short_tab2(const DevModel &m, double su,
                                                double cu, ShortTab2T<R> &st)
{
    st.s0 = (R)fma(su, m.var_cos[0], cu * m.var_sin[0]);
    st.c0 = (R)fma(cu, m.var_cos[0], -(su * m.var_sin[0]));
    st.s3 = (R)fma(su, m.var_cos[3], cu * m.var_sin[3]);
    st.c3 = (R)fma(cu, m.var_cos[3], -(su * m.var_sin[3]));
}

template <typename R, bool EN = true>
__device__ __forceinline__ void pair_core2(const PairConstsT<R> &m,
                                           const PTabT<R> &a,
                                           const ShortTab2T<R> &sa, R aks,
                                           R akc, const PTabT<R> &b, bool live,
                                           R &q, R &Yout, bool &isshort,
                                           unsigned long long &shortmask)
{
    R S = a.s * b.c - a.c * b.s;     // sin(pi (z_a - z_b) / L)
    R X = 0;
    if (EN) X = akc * b.c + aks * b.s;   // a_long * cos(...)
    R Y = S;
    bool wrapped = false;
    if (EN) wrapped = X < (R)0;      // |z_a - z_b| > L/2
    const bool neg = S < (R)0;       // sgn(D) = sgn(sin(pi D / L))
    // min-image r < rm.  (`live` = both particles exist: the idle slots of a
    // padded shape hold z = 0 and would otherwise pull whole wavefronts into
    // the generic branch whenever their partner is near the box boundary)
    isshort = live & (q_abs(S) < m.sin_rm);
    shortmask = __ballot(isshort);
    if (isshort) {
        if (!EN) wrapped = (a.c * b.c + a.s * b.s) < (R)0;
        if (wrapped == neg) {
            asm volatile("");
            const R os = neg ? sa.s3 : sa.s0, oc = neg ? sa.c3 : sa.c0;
            Y = oc * b.cu + os * b.su;
            if (EN) {
                const R xs = os * b.cu - oc * b.su;
                X = m.m_k2 * xs;
            }
        } else {
            asm volatile("");
            short_generic<R, EN>(m, a, b, S, wrapped, X, Y);
        }
    }
    if (EN) q = pair_div(X, Y);
    Yout = Y;
}

// Short-range pair, one-case form (P = 1, doubled tables).  The second copy of
// the LDS tables is what a lane reads when its partner index wraps around the
// lane group (gl - k < 0).  With position-sorted lanes that partner sits at the
// far end of the box, one period above: the copy therefore holds the tables of
// z - L (sin, cos of pi z / L negated; k2 z rotated by -k2 L), so that the
// pair looks like an ordinary one with 0 < D < L/2.  Then EVERY short-range
// pair of a sorted walker is the case (no wrap, D > 0) of pair_core4: one
// four-instruction body for the whole wave (the four-case form runs two
// bodies in every step, lanes gl >= k in one, lanes gl < k in another), no
// multiply by -k2 (folded into the own table) and 8 instead of 28 registers
// of own tables.  Lanes whose pair is not in that case (imperfect order
// after a move) take the generic path below; it is exact for any order.
template <typename R>
struct OwnShort1T {
    R s0, c0;      // sin, cos(k2 z - phi)
    R ks0, kc0;    // -k2 times the same
};

template <typename R, bool EN = true>
__device__ __forceinline__ void pair_core1(const PairConstsT<R> &m, R as,
                                           R ac, const OwnShort1T<R> &o,
                                           R aks, R akc, const PTabT<R> &b,
                                           bool lower, R own_su, R own_cu,
                                           bool live, R &q, R &Yout,
                                           bool &isshort,
                                           unsigned long long &shortmask)
{
    R S = as * b.c - ac * b.s;       // sin(pi (z_a - z_b') / L)
    R X = 0;
    if (EN) X = akc * b.c + aks * b.s;   // a_long * cos(...)
    R Y = S;
    isshort = live & (q_abs(S) < m.sin_rm);   // min-image r < rm
    shortmask = __ballot(isshort);
    if (isshort) {
        // (without the quotient only the sign of the cosine is needed, and
        // only here)
        if (!EN) X = ac * b.c + as * b.s;
        if ((S > (R)0) & (X > (R)0)) {
            // 0 < D' < L/2: theta = k2 D' - phi, X = -k2 sin, Y = cos
            asm volatile("");
            if (EN) X = o.ks0 * b.cu - o.kc0 * b.su;
            Y = o.c0 * b.cu + o.s0 * b.su;
        } else {
            asm volatile("");
            // generic (exact for any order of the lanes): undo the copy's
            // shift by one period, then the sequence of pair_core
            R bsu = b.su, bcu = b.cu, Sg = S, Xg = X;
            if (lower) {
                const R s2 = bsu * m.cth + bcu * m.sth_signed;
                const R c2 = bcu * m.cth - bsu * m.sth_signed;
                bsu = s2; bcu = c2; Sg = -S; Xg = -X;
            }
            R Su = own_su * bcu - own_cu * bsu;   // sin(k2 (z_a - z_b))
            R Cu = own_cu * bcu + own_su * bsu;
            if (Xg < (R)0) {
                // |D| > L/2: min image d = D - sgn(D) L, sgn(D) = sgn(Sg)
                const R t = (Sg < (R)0) ? -m.sth_signed : m.sth_signed;
                const R ns = Su * m.cth - Cu * t;
                const R nc = Cu * m.cth + Su * t;
                Su = ns; Cu = nc;
            }
            if (EN) {
                const R t2 = q_copysign(m.k2sphi, Su);
                X = q_fma(m.m_k2cphi, Su, Cu * t2);
            }
            Y = q_fma(q_abs(Su), m.sphi, Cu * m.cphi);
        }
    }
    if (EN) q = pair_div(X, Y);
    Yout = Y;
}

// LDS table of one lane group: 4 (5 with ZCLASS: + positions) arrays of
// DUP*G*P doubles.  For P = 1 every entry is stored twice (lane g at g and
// G + g) so a rotated read (g - k) never needs a modulo; for P >= 2 the copy
// costs occupancy through LDS (P = 8: 128 KB per block, one wave per SIMD), so
// the table is stored once and the rotated index is masked (one v_and per
// partner table, i.e. per 2-8 pairs).
#ifndef QMC_LINEAR_ORDER
#define QMC_LINEAR_ORDER 1
#endif
#ifndef QMC_LEAD_SHORT
#define QMC_LEAD_SHORT 1
#endif
#ifndef QMC_ROLLED_LOOP
#define QMC_ROLLED_LOOP 1
#endif
#ifndef QMC_TWOCASE
#define QMC_TWOCASE 1
#endif

// Tile-sweep knobs of the N = 512 shape (BASELINE.json configs[4]: "LDS
// tile-size sweep"; tools/tile_sweep.sh builds the variants): own particles per
// rotation pass (the register tile: 64 * QMC_PA8 particles) and copies of the
// LDS tables.  Defaults = the fastest measured (profiles/r02_n512_tile_sweep.txt).
#ifndef QMC_PA8
#define QMC_PA8 8
#endif
#ifndef QMC_DUP8
#define QMC_DUP8 1
#endif

#ifndef QMC_SORTED64
#define QMC_SORTED64 1
#endif
template <int G, int P, bool ZCLASS>
struct GroupLds {
    static constexpr int DUP = (P >= 8) ? QMC_DUP8 : ((P >= 2) ? 1 : 2);
    static constexpr int ROW = DUP * G * P;
    // (the stepping kernels allocate the larger of this layout and the
    // sorted-row one, qmc_kernels.h: StepLds)
    static constexpr int DOUBLES = (ZCLASS ? 5 : 4) * ROW;
};

// Evaluate one walker held in registers.
//   z[P]      : positions owned by this lane (particle index gl + G*a), inside
//               [0, L): what the pair tables are built from
//   z1[P]     : the same particles as the one-body factor sees them -- the
//               positions as the caller gave them.  The reference's pair
//               distances are minimum images (periodic in L) but its one-body
//               factor takes z mod 1 of the raw position (mrbp_qmc/model.py:
//               417, 440): in a supercell that is not a whole number of lattice
//               periods a particle outside the box does not see what its image
//               inside sees.  Everywhere but the evaluation of a
//               caller-supplied configuration z1 is z itself (same registers).
//   F[P]      : out, drift of the own particles
//   eith[P]   : out if ITH, local energy per particle
//   E         : out if EN, local energy of the walker (same value in every lane)
//   logwf     : out if WF, log|psi| (same value in every lane)
// EN = false is the log|psi|-only pass of the VMC step (no quotients, no
// drift, no energy: the Metropolis test needs none of them, and the reference
// evaluates the energy of accepted moves only, qmc_base/jastrow/vmc.py:253-262);
// REUSE = the pair tables of this configuration are already in LDS (the energy
// pass after an accepted move).
template <int G, int P, bool PAD, bool WF, bool ITH, bool ZCLASS,
          typename R = double, bool EN = true, bool REUSE = false>
__device__ __forceinline__ void eval_walker(const DevModel &m,
                                            const double (&z)[P],
                                            const double (&z1)[P], int gl,
                                            double *lds, double (&F)[P],
                                            double (&eith)[P], double &E,
                                            double &logwf)
{
    constexpr int DUP = GroupLds<G, P, ZCLASS>::DUP;
    constexpr int ROW = GroupLds<G, P, ZCLASS>::ROW;
    // (section ids of the energy pass after an accepted VMC move: second half)
    [[maybe_unused]] constexpr int QMC_SEC_OFF = REUSE ? QMC_NSEC / 2 : 0;
    // Own particles are processed PA at a time: with P = 8 the tables of all
    // eight (96 VGPRs) would leave one wave per SIMD, so the rotation runs in
    // two passes of four own particles (tables re-read from LDS).
    constexpr int PA = (P > 4) ? QMC_PA8 : P;
    constexpr int NPASS = P / PA;
    constexpr bool RD = sizeof(R) == 8;      // the pair loop runs in double
    R *lS = (R *)lds, *lC = lS + ROW, *lSU = lS + 2 * ROW,
      *lCU = lS + 3 * ROW, *lZ = lS + 4 * ROW;
    const int n = m.n;
    const int ge = lanes_in_use<G, PAD>(m);   // lanes the rotation runs over
    // one-case form with a shifted second copy of the tables (pair_core1)
    // (one particle per lane; in double only where the kernels keep the lanes
    // in ascending position, i.e. one walker per wavefront: LINEAR_ORDER)
    constexpr bool ROTCOPY = (P == 1) && (DUP == 2) && !ZCLASS &&
                             (!RD || (G == 64 && QMC_LINEAR_ORDER));
    // two particles per lane in ascending rows: two-case form (pair_core2)
    constexpr bool TWOCASE = QMC_TWOCASE && (P == 2) && (G == 64) && !ZCLASS &&
                             QMC_LINEAR_ORDER && SlotMap<G, P>::CONSECUTIVE;
    // four-case short-range form while the own tables fit (see pair_core4)
    constexpr bool FOURCASE = (P <= 2) && !ROTCOPY && !TWOCASE;
    PTabT<R> t[PA];
    ShortTabT<R> st4[FOURCASE ? PA : 1];
    OwnShort1T<R> os1[ROTCOPY ? PA : 1];
    ShortTab2T<R> st2[TWOCASE ? PA : 1];
    R aks[PA], akc[PA];      // a_long * (sin, cos)(pi z / L)
    const R a_long_r = (R)m.a_long;
    bool ok[P];
    double kin1[P];          // one-body kinetic + potential (ITH)
    double kin1_sum = 0.0;   // their sum over the own particles (!ITH)
    R prodS = 1, prodL = 1;  // running products of pair factors (WF)
    double prod1 = 1.0;      // product of the one-body factors (WF)
    double xoff_sum = 0.0;   // sum of the exponents split off them (one_body)
    int expS = 0, expL = 0;  // binary exponents split off the products
    int exp1 = 0;            // ... and off the one-body product (P >= 4)
    int nshort = 0, npair = 0;
    // one walker per wavefront and no padding: short pairs are counted with a
    // ballot + scalar popcount (SALU) instead of a per-lane VALU add
    constexpr bool WAVE_COUNT = (G == 64) && !PAD;
    int ns_wave = 0;
    int nb_wave = 0;         // particles inside a barrier (one-body table path)
    bool nb_counted = false;
    R Qall = 0, Qs = 0;      // sum of q^2 over all / short pairs
    R Kown[P], KT[P];        // per-particle pair kinetic sums (ITH)
    R T[P];                  // travelling drift of the partner lane
    R Fr[P];                 // drift sums of the pair loop

    QMC_SECTION("tables+onebody");
#pragma unroll
    for (int a = 0; a < P; ++a) {
        ok[a] = !PAD || lane_particle<G, P, PAD>(m, gl, a) < n;
        F[a] = 0.0; T[a] = 0; Kown[a] = 0; KT[a] = 0;
        if (ITH) kin1[a] = 0.0;
        // (the row of the pair-table angles is requested first: 10 registers
        // wait for it while the one-body factor is evaluated)
        TrigRow trow;
        const bool trig_ok = !REUSE && !m.is_ideal && m.trig_table &&
                             trig_tab_load(m, z[a], trow);
        // the one-body factor first: its table rows (or its transcendental
        // branches) are done with before the pair tables occupy registers
        if (!m.is_free && m.ob_table) {
            double ldz = 0.0, lf = 0.0;
            bool barrier;
            one_body_tab<WF, EN>(m, z1[a], ldz, lf, barrier);
            if (!EN) {
                if (WF && ok[a]) xoff_sum -= lf;
            } else if (WAVE_COUNT && !ITH && m.uniform_barrier) {
                // one walker per wavefront, every barrier alike: the region
                // constants are counted on the scalar unit and added once
                nb_wave += __popcll(__ballot(barrier));
                nb_counted = true;
                F[a] = ldz;
                kin1_sum = fma(ldz, ldz, kin1_sum);
                if (WF) xoff_sum -= lf;
            } else if (ok[a]) {
                const double kp =
                    fma(ldz, ldz, one_body_kin_const(m, z1[a], barrier));
                F[a] = ldz;
                if (ITH) kin1[a] = kp; else kin1_sum += kp;
                // log f1 joins the sum of split-off exponents (subtracted)
                if (WF) xoff_sum -= lf;
            }
        } else if (!m.is_free) {
            double ldz, kp, f1, xoff;
            one_body(m, z1[a], ldz, kp, f1, xoff);
            if (ok[a]) {
                if (EN) {
                    F[a] = ldz;
                    if (ITH) kin1[a] = kp; else kin1_sum += kp;
                }
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
        // (in double the pair sums continue from the one-body term, as ever)
        Fr[a] = RD ? (R)F[a] : (R)0;
        __builtin_amdgcn_sched_barrier(0);
        if (!m.is_ideal) {
            PTab ta;
            int i0 = a * DUP * G + gl;
            if (REUSE) {
                // the entry this lane published in the first pass (ROTCOPY:
                // the upper copy is the particle itself)
                const int io = ROTCOPY ? i0 + ge : i0;
                ta.s = (double)lS[io]; ta.c = (double)lC[io];
                ta.su = (double)lSU[io]; ta.cu = (double)lCU[io];
            } else if (trig_ok) {
                trig_tab_finish(m, trow, ta);
            } else {
                sincos_halfpi(z[a] * m.two_over_L, ta.s, ta.c);
                sincos_halfpi(z[a] * m.k2_2pi, ta.su, ta.cu);
            }
            if (NPASS == 1) {
                t[a % PA].s = (R)ta.s; t[a % PA].c = (R)ta.c;
                t[a % PA].su = (R)ta.su; t[a % PA].cu = (R)ta.cu;
                if (EN) {
                    aks[a % PA] = (R)(m.a_long * ta.s);
                    akc[a % PA] = (R)(m.a_long * ta.c);
                } else {
                    aks[a % PA] = 0; akc[a % PA] = 0;
                }
                if (FOURCASE) make_short_tab<R>(m, ta.su, ta.cu, st4[a % PA]);
                if (TWOCASE) make_short_tab2<R>(m, ta.su, ta.cu, st2[a % PA]);
                if (ROTCOPY) {
                    OwnShort1T<R> &o = os1[a % PA];
                    const double s0 = fma(ta.su, m.cphi, -(ta.cu * m.sphi));
                    const double c0 = fma(ta.cu, m.cphi, ta.su * m.sphi);
                    o.s0 = (R)s0; o.c0 = (R)c0;
                    o.ks0 = (R)(m.m_k2 * s0);
                    o.kc0 = (R)(m.m_k2 * c0);
                }
            }
            // (idle lanes of a padded shape must not write: with two copies
            // their slot is another lane's)
            if (REUSE || (PAD && gl >= ge)) {
            } else if (ROTCOPY) {
                // upper copy: the particle itself; lower copy (read when the
                // partner index wraps): the particle one period below
                lS[i0 + ge] = (R)ta.s; lC[i0 + ge] = (R)ta.c;
                lSU[i0 + ge] = (R)ta.su; lCU[i0 + ge] = (R)ta.cu;
                lS[i0] = (R)-ta.s; lC[i0] = (R)-ta.c;
                lSU[i0] = (R)fma(ta.su, m.cth, -(ta.cu * m.sth_signed));
                lCU[i0] = (R)fma(ta.cu, m.cth, ta.su * m.sth_signed);
            } else {
                lS[i0] = (R)ta.s; lC[i0] = (R)ta.c;
                lSU[i0] = (R)ta.su; lCU[i0] = (R)ta.cu;
                if (ZCLASS) lZ[i0] = (R)z[a];
                if (DUP == 2) {
                    lS[i0 + ge] = (R)ta.s; lC[i0 + ge] = (R)ta.c;
                    lSU[i0 + ge] = (R)ta.su; lCU[i0 + ge] = (R)ta.cu;
                    if (ZCLASS) lZ[i0 + ge] = (R)z[a];
                }
            }
        } else if (NPASS == 1) {
            aks[a % PA] = 0; akc[a % PA] = 0;
        }
    }

    // Split the binary exponent off a running product so that it can neither
    // underflow nor overflow (one frexp pair instead of a log per fold).
#define QMC_FOLD(p, e) q_fold(p, e)
    // own table of particle (lane gl, register a) back from LDS
#define QMC_LOAD_OWN(dst, a)                                                  \
    do {                                                                      \
        const int i0_ = (a) * DUP * G + gl;                                   \
        (dst).s = lS[i0_]; (dst).c = lC[i0_];                                 \
        (dst).su = lSU[i0_]; (dst).cu = lCU[i0_];                             \
    } while (0)

    if (!m.is_ideal) {
        const PairConstsT<R> pc = load_pair_consts<R>(m);
        // make the table visible to the other lanes of the wave (one wave owns
        // its groups' LDS region: LDS ops of a wave complete in order)
        if (!REUSE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }

        // sums that count every unordered pair once
#define QMC_TALLY(q, Y, isshort)                                              \
    do {                                                                      \
        if (EN) Qall = q_fma(q, q, Qall);                                     \
        if (!WAVE_COUNT) ++npair;                                             \
        /* prodL runs over ALL pairs (no else branch, no second compare);   \
           the short factors are divided out once at the end */             \
        if (WF) prodL *= Y;          /* sign dropped at the end */            \
        if (isshort) {                                                        \
            asm volatile("");   /* exec-masked, not selects */                \
            if (EN) Qs = q_fma(q, q, Qs);                                     \
            if (!WAVE_COUNT) ++nshort;                                        \
            if (WF) prodS *= Y;                                               \
        }                                                                     \
    } while (0)
#define QMC_PAIR_KIN(q, isshort)                                              \
    ((isshort) ? q_fma(q, q, pc.k2sq)                                         \
               : q_fma((q) * (q), pc.inv_beta, pc.b_long))

        // ---- k = 0: pairs inside the lane ----
        QMC_SECTION("pairs_in_lane");
#pragma unroll
        for (int a = 0; a < P; ++a) {
            PTabT<R> ta; R aksa, akca;
            if (NPASS == 1) {
                ta = t[a % PA]; aksa = aks[a % PA]; akca = akc[a % PA];
            } else if (a + 1 < P) {
                QMC_LOAD_OWN(ta, a);
                aksa = a_long_r * ta.s; akca = a_long_r * ta.c;
            }
#pragma unroll
            for (int b = a + 1; b < P; ++b) {
                PTabT<R> tb;
                if (NPASS == 1) tb = t[b % PA];
                else QMC_LOAD_OWN(tb, b);
                R q = 0, Y; bool sh; unsigned long long shm;
                if (FOURCASE)
                    pair_core4<ZCLASS, R, EN>(pc, ta, st4[a % PA], aksa, akca,
                                              (R)z[a], tb, (R)z[b],
                                              !PAD || (ok[a] && ok[b]), q, Y,
                                              sh, shm);
                else
                    pair_core<ZCLASS, R, EN>(pc, ta, aksa, akca, (R)z[a], tb,
                                             (R)z[b], !PAD || (ok[a] && ok[b]),
                                             q, Y, sh, shm);
                if (WAVE_COUNT)
                    ns_wave += __popcll(shm);
                if (!PAD || (ok[a] && ok[b])) {
                    if (EN) { Fr[a] += q; Fr[b] -= q; }
                    QMC_TALLY(q, Y, sh);
                    if (ITH) {
                        R kk = QMC_PAIR_KIN(q, sh);
                        Kown[a] += kk; Kown[b] += kk;
                    }
                }
            }
            if (WF && P > 4) { QMC_FOLD(prodS, expS); QMC_FOLD(prodL, expL); }
        }

        // ---- k = 1 .. G/2: rotate over partner lanes ----
        constexpr bool ROT_DPP = QMC_T_DPP && (G == 64 || G == 16) && !PAD;
        const int lane = threadIdx.x & 63;
        // the lane below in the ring of the ge lanes in use
        const int src = lane - gl + (PAD ? (gl == 0 ? ge - 1 : gl - 1)
                                         : ((gl + G - 1) & (G - 1)));
        // ... and the lane half a ring away (delivery of the travelling sums)
        int half_src = gl + ge / 2;
        if (half_src >= ge) half_src -= ge;
        half_src += lane - gl;
        // One rotation step of pass H (own particles H*PA .. H*PA+PA-1);
        // LAST is a compile-time flag: the final half step (k = G/2) visits
        // every pair from both sides, so each side only updates its own
        // particle and the lower half of the lanes tallies.
#define QMC_KSTEP(H, k, LAST)                                                 \
        {                                                                     \
            const bool count_pair = !(LAST) || gl < ge / 2;                   \
            /* partner-major order: one partner table live at a time */       \
            _Pragma("unroll")                                                 \
            for (int b = 0; b < P; ++b) {                                     \
                PTabT<R> pb;                                                  \
                int pl = gl - (k);                                            \
                if (PAD) { if (pl < 0) pl += ge; } else pl &= G - 1;          \
                const int idx = (DUP == 2) ? b * 2 * G + gl + ge - (k)        \
                                           : b * G + pl;                      \
                pb.s = lS[idx]; pb.c = lC[idx];                               \
                pb.su = lSU[idx]; pb.cu = lCU[idx];                           \
                const R pz = ZCLASS ? lZ[idx] : (R)0;                         \
                const bool pok = !PAD ||                                      \
                    (gl < ge && lane_particle<G, P, PAD>(m, pl, b) < n);      \
                _Pragma("unroll")                                             \
                for (int a = 0; a < PA; ++a) {                                \
                    constexpr int ao_base = (H) * PA;                         \
                    R q = 0, Y; bool sh; unsigned long long shm;             \
                    const bool live = !PAD || (ok[ao_base + a] && pok);       \
                    if (ROTCOPY)                                              \
                        pair_core1<R, EN>(pc, t[a].s, t[a].c,                 \
                                      os1[ROTCOPY ? a : 0], aks[a], akc[a],   \
                                      pb, gl < (k),                           \
                                      lSU[a * DUP * G + ge + gl],             \
                                      lCU[a * DUP * G + ge + gl], live, q, Y, \
                                      sh, shm);                               \
                    else if (TWOCASE)                                         \
                        pair_core2<R, EN>(pc, t[a], st2[TWOCASE ? a : 0],     \
                                      aks[a], akc[a], pb, live, q, Y, sh,     \
                                      shm);                                   \
                    else if (FOURCASE)                                        \
                        pair_core4<ZCLASS, R, EN>(pc, t[a],                   \
                                              st4[FOURCASE ? a : 0], aks[a],  \
                                              akc[a], (R)z[ao_base + a], pb,  \
                                              pz, live, q, Y, sh, shm);       \
                    else                                                      \
                        pair_core<ZCLASS, R, EN>(pc, t[a], aks[a], akc[a],    \
                                             (R)z[ao_base + a], pb, pz, live, \
                                             q, Y, sh, shm);                  \
                    /* G = 64: the lower half of the lanes is bits 0..31 */   \
                    if (WAVE_COUNT)                                           \
                        ns_wave += __popcll((LAST) ? (shm & 0xffffffffull)    \
                                                   : shm);                    \
                    if (live) {                                               \
                        if (EN) Fr[ao_base + a] += q;                         \
                        if (EN && !(LAST)) T[b] -= q;                         \
                        if (!(LAST) || count_pair) { QMC_TALLY(q, Y, sh); }   \
                        if (ITH) {                                            \
                            R kk = QMC_PAIR_KIN(q, sh);                       \
                            Kown[ao_base + a] += kk;                          \
                            if (!(LAST)) KT[b] += kk;                         \
                        }                                                     \
                    }                                                         \
                }                                                             \
                /* large P: keep the scheduler from interleaving the pairs   \
                   of different partners (it trades the occupancy away       \
                   for it: 282 registers instead of ~180 at P = 8) */        \
                if (P >= 4) __builtin_amdgcn_sched_barrier(0);                \
                /* float products: the P pairs of one partner are all near   \
                   or all far (consecutive slots); P^2 factors of a few     \
                   per cent would leave the float range before the fold at  \
                   the end of the step */                                    \
                if (WF && !RD && P >= 4) {                                    \
                    QMC_FOLD(prodS, expS);                                    \
                    QMC_FOLD(prodL, expL);                                    \
                }                                                             \
            }                                                                 \
            if (EN && !(LAST)) {                                              \
                _Pragma("unroll")                                             \
                for (int b = 0; b < P; ++b) {                                 \
                    if constexpr (ROT_DPP) {                                  \
                        T[b] = group_ror1<G>(T[b]);                           \
                        if (ITH) KT[b] = group_ror1<G>(KT[b]);                \
                    } else {                                                  \
                        T[b] = __shfl(T[b], src, 64);                         \
                        if (ITH) KT[b] = __shfl(KT[b], src, 64);              \
                    }                                                         \
                }                                                             \
            }                                                                 \
            /* P = 1: 16 factors between folds, each >= sin(pi rm / L) or   \
               cos(k2 rm - phi): no underflow for any admissible model */     \
            if (WF && (((k) & 15) == 0 || P > 1)) {                           \
                QMC_FOLD(prodS, expS);                                        \
                QMC_FOLD(prodL, expL);                                        \
            }                                                                 \
        }
#if QMC_ROLLED_LOOP
#define QMC_ROLL_PRAGMA _Pragma("clang loop unroll(disable)")
#else
#define QMC_ROLL_PRAGMA
#endif
#define QMC_PASS(H)                                                           \
        if ((H) < NPASS) {                                                    \
            if (NPASS > 1) {                                                  \
                _Pragma("unroll")                                             \
                for (int a = 0; a < PA; ++a) {                                \
                    QMC_LOAD_OWN(t[a], (H) * PA + a);                         \
                    aks[a] = a_long_r * t[a].s;                               \
                    akc[a] = a_long_r * t[a].c;                               \
                }                                                             \
            }                                                                 \
            /* (kept rolled: unrolled, the scheduler hoists the LDS reads of \
               every copy and the kernel loses half its occupancy: -8 %) */  \
            QMC_SECTION("rotation_loop_body");                                \
            QMC_ROLL_PRAGMA                                                   \
            for (int k = ((H) == 0 ? k_first : 1); k < ge / 2; ++k)           \
                QMC_KSTEP(H, k, false)                                        \
            QMC_SECTION("rotation_last_step");                                \
            QMC_KSTEP(H, ge / 2, true)                                        \
            /* deliver the travelling sums to their owners (lane gl ^ G/2    \
               holds them) and start the next pass from zero */              \
            _Pragma("unroll")                                                 \
            for (int b = 0; EN && b < P; ++b) {                               \
                Fr[b] += PAD ? __shfl(T[b], half_src, 64)                     \
                             : __shfl_xor(T[b], G / 2, 64);                   \
                T[b] = 0;                                                     \
                if (ITH) {                                                    \
                    Kown[b] += PAD ? __shfl(KT[b], half_src, 64)              \
                                   : __shfl_xor(KT[b], G / 2, 64);            \
                    KT[b] = 0;                                                \
                }                                                             \
            }                                                                 \
        }
        // One walker per wavefront in ascending order: the first rotation
        // steps (partners a few lanes away) are short-range pairs for EVERY
        // lane -- about the first third of the steps at rm = L/4 -- and need
        // only the partner's k2-table, no classification by sin(pi d / L) and
        // no exec-masked region.  k = 1, 2 (neighbours: between two passes of
        // the transposition sort some are out of order, which sends the whole
        // wave through the generic branch of pair_core1 in four steps of five)
        // use the form that is exact for either sign of the separation: 10
        // vector instructions + the division against ~40.  From k = 3 the pairs
        // are the single case of the shifted tables: X, Y of the one-case form
        // and three compares that decide exactly whether the pair is short,
        // unwrapped and ordered (DevModel.sp_*), 18 instructions per step
        // against 24.  The first step where some lane says no is redone by the
        // general loop, which takes over from there.
        int k_first = 1;
        constexpr bool LEAD_SHORT = QMC_LEAD_SHORT && ROTCOPY && (G == 64) &&
                                    !PAD && (P == 1);
        if constexpr (LEAD_SHORT) {
            if (pc.sp_ok) {
                R Q1 = 0, P1 = 1;          // tallies of these steps
                int e1 = 0;
                bool lead = true;
#define QMC_LEAD_TAIL(X, Y)                                                   \
                {                                                             \
                    ns_wave += 64;                                            \
                    if (WF) P1 *= Y;                                          \
                    if (EN) {                                                 \
                        const R q = pair_div(X, Y);                           \
                        Fr[0] += q;                                           \
                        T[0] -= q;                                            \
                        Q1 = q_fma(q, q, Q1);                                 \
                        if (ITH) {                                            \
                            const R kk = q_fma(q, q, pc.k2sq);                \
                            Kown[0] += kk;                                    \
                            KT[0] += kk;                                      \
                        }                                                     \
                        T[0] = group_ror1<G>(T[0]);                           \
                        if (ITH) KT[0] = group_ror1<G>(KT[0]);                \
                    }                                                         \
                }
                QMC_SECTION("leading_neighbour_steps");
                {
                    const R osu = lSU[G + gl], ocu = lCU[G + gl];
                    for (; k_first <= 2; ++k_first) {
                        const int idx = gl + G - k_first;
                        const R bsu = lSU[idx], bcu = lCU[idx];
                        const R Su = osu * bcu - ocu * bsu;   // sin(k2 D')
                        const R Cu = ocu * bcu + osu * bsu;   // cos(k2 D')
                        // |k2 D'| < pi: |D'| < rm iff cos(k2 D') > cos(k2 rm)
                        if (__builtin_amdgcn_ballot_w64(Cu > pc.sp_cos) != ~0ull) {
                            lead = false;
                            break;
                        }
                        R Xs = 0;
                        if (EN) {
                            const R t2 = q_copysign(pc.k2sphi, Su);
                            Xs = q_fma(pc.m_k2cphi, Su, Cu * t2);
                        }
                        const R Ys = q_fma(q_abs(Su), pc.sphi, Cu * pc.cphi);
                        QMC_LEAD_TAIL(Xs, Ys)
                    }
                }
                QMC_SECTION("leading_short_steps");
                if (lead) {
                    const OwnShort1T<R> o = os1[0];
                    for (; k_first < G / 2; ++k_first) {
                        const int idx = gl + G - k_first;
                        const R bsu = lSU[idx], bcu = lCU[idx];
                        const R Xs = o.ks0 * bcu - o.kc0 * bsu;
                        const R Ys = o.c0 * bcu + o.s0 * bsu;
                        // (one ballot per compare: each is the compare's own
                        // mask; all 64 lanes are active here)
                        const unsigned long long fast =
                            __builtin_amdgcn_ballot_w64(Xs > pc.sp_xlo) &
                            __builtin_amdgcn_ballot_w64(Xs < pc.sp_xhi) &
                            __builtin_amdgcn_ballot_w64(Ys > (R)0);
                        if (fast != ~0ull) break;
                        QMC_LEAD_TAIL(Xs, Ys)
                        if (WF && (k_first & 15) == 0) {
                            asm volatile("");   // a branch, not selects
                            QMC_FOLD(P1, e1);
                        }
                    }
                }
#undef QMC_LEAD_TAIL
                // these pairs belong to the tallies of all pairs and of the
                // short ones
                if (EN) { Qall += Q1; Qs += Q1; }
                if (WF) {
                    QMC_FOLD(P1, e1);
                    prodS *= P1; prodL *= P1;
                    expS += e1; expL += e1;
                }
            }
        }
        QMC_SECTION("rotation");
        QMC_PASS(0)
        QMC_PASS(1)
        QMC_PASS(2)
        QMC_PASS(3)
#undef QMC_PASS
#undef QMC_KSTEP
    }
#pragma unroll
    for (int a = 0; EN && a < P; ++a)
        F[a] = RD ? (double)Fr[a] : F[a] + (double)Fr[a];

    // ---- local energy ----
    QMC_SECTION("energy+logwf");
    double e_lane = 0.0;
    if (!EN) {
    } else if (ITH) {
#pragma unroll
        for (int a = 0; a < P; ++a) {
            double e = ok[a] ? ((double)Kown[a] + kin1[a] - F[a] * F[a]) : 0.0;
            eith[a] = e;
            e_lane += e;
        }
    } else {
        // sum over unordered pairs of (k2^2 + q^2) [short] and
        // (b_long + q^2 / beta) [long], counted for both partners
        int nlong = npair - nshort;
        const double Qall_d = (double)Qall, Qs_d = (double)Qs;
        double pk = Qs_d + (Qall_d - Qs_d) * m.inv_beta;
        if (!WAVE_COUNT)
            pk += m.k2sq * (double)nshort + m.b_long * (double)nlong;
        e_lane = 2.0 * pk;
        e_lane += kin1_sum;
#pragma unroll
        for (int a = 0; a < P; ++a)
            if (ok[a]) e_lane -= F[a] * F[a];
    }
    // (one walker per wavefront: the sums over the lanes run on the matrix
    // cores, the energy's together with log|psi|'s further down)
    // (one particle per lane only: with more, the accumulator registers of the
    // matrix instruction cost the N = 128 DMC step a wave of occupancy, -5 %)
    constexpr bool MSUM = QMC_MFMA_SUM && (G == 64) && (P == 1);
    if (!EN) {
    } else if (!MSUM) E = group_sum<G>(e_lane);
    else if (!WF) E = wave_sum_mfma(e_lane);
    double e_consts = 0.0;
    if (!EN) {
    } else if (WAVE_COUNT && !ITH && nb_counted) {
        // one-body region constants of the n particles, nb_wave in a barrier
        e_consts += (double)(n - nb_wave) * m.e0 +
                    (double)nb_wave * (m.v_barrier - m.v0_minus_e0);
    }
    if (EN && WAVE_COUNT && !ITH && !m.is_ideal) {
        int nl_wave = n * (n - 1) / 2 - ns_wave;
        e_consts += 2.0 * (m.k2sq * (double)ns_wave + m.b_long * (double)nl_wave);
    }
    if (EN && !(MSUM && WF)) E += e_consts;
    if (WF) {
        const double LN2 = 0.693147180559945309417;
        // prodL holds every pair's |Y|, prodS the short ones (cos > 0):
        // long product = prodL / prodS (mantissas; exponents kept apart)
        const double prodS_d = (double)prodS, prodL_d = (double)prodL;
        // log(prod1 prodS) + beta log|prodL / prodS| as two logarithms and no
        // division (from the table the one-body factors arrive as logarithms:
        // prod1 = 1)
        double lw;
        if (!m.is_free && m.ob_table) {
            const double lS = log_pos(prodS_d);
            lw = fma(m.beta, log_pos(fabs(prodL_d)) - lS, lS);
        } else {
            lw = log_pos(prod1 * prodS_d) +
                 m.beta * log_pos(fabs(fast_div(prodL_d, prodS_d)));
        }
        lw += LN2 * ((double)(expS + exp1) + m.beta * (double)(expL - expS)) -
              xoff_sum;
        if (!WAVE_COUNT) lw += (double)nshort * m.log_am;
        if (MSUM && EN) {
            wave_sum2_mfma(e_lane, lw, E, logwf);
            E += e_consts;
        } else if (MSUM) {
            logwf = wave_sum_mfma(lw);
        } else {
            logwf = group_sum<G>(lw);
        }
        if (WAVE_COUNT) logwf += (double)ns_wave * m.log_am;
    }
#undef QMC_FOLD
#undef QMC_LOAD_OWN
#undef QMC_TALLY
#undef QMC_PAIR_KIN
}
