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
// identities give sin/cos of a_i-a_j and b_i-b_j with 4 FMA-class ops each;
//   long range  (r >= rm): f2'/f2 sgn = (pi/L) beta cot(a_i-a_j)       (period L:
//                          the minimum image needs no explicit wrap)
//   short range (r <  rm): f2'/f2     = -k2 tan(k2 r - k2 r_off)
// both reduce to ONE division q = X/Y per pair, and
//   -f2''/f2 + (f2'/f2)^2 = c_B (1 + q^2),  c_B = k2^2 or (pi/L)^2 beta.
// (reference formulas: mrbp_qmc/model.py:468-529).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define QMC_PI 3.141592653589793238462643383279502884

struct DevModel {
    int n;                 // boson_number
    int is_free, is_ideal;
    int defects_sep;
    int zclass;            // classify pairs from positions (rm close to L/2)
    double L, half_L, rm, L_minus_rm;
    double pi_L;           // pi / L
    double k2, k2sq;
    double cphi, sphi;     // cos/sin(k2 r_off)
    double cth, sth;       // cos/sin(k2 L)
    double sin_rm;         // sin(pi rm / L)
    double a_long, b_long; // (pi/L) beta, (pi/L)^2 beta
    double beta, log_am;
    // one-body (Kronig-Penney)
    double z_a, z_b, k1, kp1, e0, v0, v0d, v0_minus_e0, cf;
};

// ---------------------------------------------------------------- RNG ----
// Philox4x32-10 (Salmon et al. 2011), counter = (slot, step, index, stream),
// key = seed.  Same algorithm as the oracle so seeded runs line up.
enum { STREAM_VMC_MOVE = 0, STREAM_VMC_ACCEPT = 1, STREAM_DMC_BRANCH = 2,
       STREAM_DMC_DIFFUSE = 3 };

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0,
                                              uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t h0 = __umulhi(M0, c[0]), l0 = M0 * c[0];
        uint32_t h1 = __umulhi(M1, c[2]), l1 = M1 * c[2];
        uint32_t n0 = h1 ^ c[1] ^ k0;
        uint32_t n2 = h0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = l1; c[2] = n2; c[3] = l0;
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

__device__ __forceinline__ double philox_normal(uint64_t seed, uint32_t slot,
                                                uint32_t step, uint32_t index,
                                                uint32_t stream)
{
    double u0, u1;
    philox_uniform2(seed, slot, step, index, stream, u0, u1);
    return sqrt(-2.0 * log(1.0 - u0)) * cos(6.283185307179586476925 * u1);
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

template <int G>
__device__ __forceinline__ int group_sum_int(int v)
{
#pragma unroll
    for (int m = 1; m < G; m <<= 1)
        v += __shfl_xor(v, m, 64);
    return v;
}

__device__ __forceinline__ double flip_sign_if(double x, bool neg)
{
    return neg ? -x : x;
}

// Per-particle table entry kept in registers by the owner and published to LDS.
struct PTab {
    double s, c;    // sin/cos(pi z / L)
    double su, cu;  // sin/cos(k2 z)
};

// One-body factor (mrbp_qmc/model.py:404-464) and lattice potential (:533-551).
// ldz  = f1'/f1; kin = -f1''/f1 + ldz^2 + V(z); f1 > 0 is the factor itself.
__device__ __forceinline__ void one_body(const DevModel &m, double z,
                                         double &ldz, double &kin_pot,
                                         double &f1)
{
    double n_cell = floor(z);
    double z_cell = z - n_cell;
    bool barrier = m.z_a < z_cell;
    if (barrier) {
        double x = m.kp1 * (z_cell - 1.0 + 0.5 * m.z_b);
        double t = tanh(x);
        ldz = m.kp1 * t;
        f1 = cosh(x);
        long long nc = (long long)n_cell;
        long long r = nc % m.defects_sep;
        if (r < 0) r += m.defects_sep;
        double v = (r == 0) ? m.v0d : m.v0;
        kin_pot = -m.v0_minus_e0 + ldz * ldz + v;
    } else {
        double x = m.k1 * (z_cell - 0.5 * m.z_a);
        double sx, cx;
        sincos(x, &sx, &cx);
        ldz = -m.k1 * (sx / cx);
        f1 = m.cf * cx;
        kin_pot = m.e0 + ldz * ldz;
    }
}

// Result of one pair evaluation, as seen from the "own" particle i.
//   w   : contribution to drift_i (partner j gets -w)
//   q   : the ratio X/Y;   isshort : r < rm
//   fac : |f2| up to the constant am (short: cos(k2 r - phi), long: |sin|)
struct PairOut {
    double w, q, fac;
    bool isshort;
};

template <bool ZCLASS>
__device__ __forceinline__ PairOut pair_eval(const DevModel &m, const PTab &a,
                                             double za, const PTab &b,
                                             double zb)
{
    PairOut o;
    double S = a.s * b.c - a.c * b.s;     // sin(pi (z_a - z_b) / L)
    double C = a.c * b.c + a.s * b.s;     // cos
    double Su = a.su * b.cu - a.cu * b.su; // sin(k2 (z_a - z_b))
    double Cu = a.cu * b.cu + a.su * b.su;
    bool neg, wrapped, isshort;
    if (ZCLASS) {
        double D = za - zb;
        double aD = fabs(D);
        neg = D < 0.0;
        wrapped = aD > m.half_L;
        isshort = (aD < m.rm) | (aD > m.L_minus_rm);
    } else {
        neg = S < 0.0;                    // sign(z_a - z_b)
        wrapped = C < 0.0;                // |z_a - z_b| > L/2
        isshort = fabs(S) < m.sin_rm;     // min-image r < rm
    }
    // sin/cos of k2 * r for the min-image distance r (valid when short)
    double Pq = flip_sign_if(Su, neg);    // sin(k2 |D|)
    double Aw = m.sth * Cu - m.cth * Pq;  // sin(k2 (L - |D|))
    double Bw = m.cth * Cu + m.sth * Pq;
    double A = wrapped ? Aw : Pq;
    double B = wrapped ? Bw : Cu;
    double Xs = A * m.cphi - B * m.sphi;  // sin(k2 r - phi)
    double Ys = B * m.cphi + A * m.sphi;  // cos(k2 r - phi)
    double X = isshort ? Xs : C;
    double Y = isshort ? Ys : S;
    double q = X / Y;
    // sign of the min-image separation d: flips when the pair wraps
    bool dneg = neg != wrapped;
    double cs = flip_sign_if(m.k2, !dneg); // -k2 * sgn(d)
    double cA = isshort ? cs : m.a_long;
    o.w = cA * q;
    o.q = q;
    o.fac = isshort ? Ys : fabs(S);
    o.isshort = isshort;
    return o;
}

// LDS table of one lane group: 4 (5 with ZCLASS: + positions) arrays of 2*G*P doubles; the
// entry of particle (lane g, register b) is stored at b*2G + g and b*2G + G + g
// so a rotated read (g - k) never needs a modulo.
template <int G, int P, bool ZCLASS>
struct GroupLds {
    static constexpr int ROW = 2 * G * P;
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
    constexpr int ROW = 2 * G * P;
    double *lS = lds, *lC = lds + ROW, *lSU = lds + 2 * ROW,
           *lCU = lds + 3 * ROW, *lZ = lds + 4 * ROW;
    const int n = m.n;
    PTab t[P];
    bool ok[P];
    double kin1[P];          // one-body kinetic + potential
    double prod = 1.0;       // running product of positive factors (WF)
    double lsum = 0.0;       // accumulated logs (WF)
    int nshort = 0, npair = 0;
    double Qs = 0.0, Ql = 0.0; // sum of q^2 over short / long pairs
    double Kown[P], KT[P];   // per-particle pair kinetic sums (ITH)
    double T[P];             // travelling drift of the partner lane

#pragma unroll
    for (int a = 0; a < P; ++a) {
        ok[a] = !PAD || (gl + G * a) < n;
        F[a] = 0.0; kin1[a] = 0.0; T[a] = 0.0; Kown[a] = 0.0; KT[a] = 0.0;
        if (!m.is_ideal) {
            sincos(z[a] * m.pi_L, &t[a].s, &t[a].c);
            sincos(z[a] * m.k2, &t[a].su, &t[a].cu);
            int i0 = a * 2 * G + gl;
            lS[i0] = t[a].s;   lS[i0 + G] = t[a].s;
            lC[i0] = t[a].c;   lC[i0 + G] = t[a].c;
            lSU[i0] = t[a].su; lSU[i0 + G] = t[a].su;
            lCU[i0] = t[a].cu; lCU[i0 + G] = t[a].cu;
            if (ZCLASS) { lZ[i0] = z[a]; lZ[i0 + G] = z[a]; }
        }
        if (!m.is_free) {
            double ldz, kp, f1;
            one_body(m, z[a], ldz, kp, f1);
            if (ok[a]) {
                F[a] = ldz;
                kin1[a] = kp;
                if (WF) prod *= f1;
            }
        }
    }
    if (WF) { lsum = log(prod); prod = 1.0; }

    if (!m.is_ideal) {
        // make the table visible to the other lanes of the wave (one wave owns
        // its groups' LDS region: LDS ops of a wave complete in order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        double prodL = 1.0;  // product of |sin| over long pairs (WF)

        // ---- k = 0: pairs inside the lane ----
#pragma unroll
        for (int a = 0; a < P; ++a) {
#pragma unroll
            for (int b = a + 1; b < P; ++b) {
                PairOut o = pair_eval<ZCLASS>(m, t[a], z[a], t[b], z[b]);
                bool v = ok[a] && ok[b];
                if (v) {
                    F[a] += o.w; F[b] -= o.w;
                    if (o.isshort) { Qs = fma(o.q, o.q, Qs); ++nshort; }
                    else           { Ql = fma(o.q, o.q, Ql); }
                    ++npair;
                    if (WF) { if (o.isshort) prod *= o.fac; else prodL *= o.fac; }
                    if (ITH) {
                        double cB = o.isshort ? m.k2sq : m.b_long;
                        double kk = cB * fma(o.q, o.q, 1.0);
                        Kown[a] += kk; Kown[b] += kk;
                    }
                }
            }
        }

        // ---- k = 1 .. G/2: rotate over partner lanes ----
        for (int k = 1; k <= G / 2; ++k) {
            const bool last = (k == G / 2);
            const bool count_pair = !last || gl < G / 2;
            PTab pb[P];
            double pz[P];
            bool pok[P];
#pragma unroll
            for (int b = 0; b < P; ++b) {
                int idx = b * 2 * G + gl + G - k;
                pb[b].s = lS[idx]; pb[b].c = lC[idx];
                pb[b].su = lSU[idx]; pb[b].cu = lCU[idx];
                pz[b] = ZCLASS ? lZ[idx] : 0.0;
                int pl = gl - k; if (pl < 0) pl += G;
                pok[b] = !PAD || (pl + G * b) < n;
            }
#pragma unroll
            for (int a = 0; a < P; ++a) {
#pragma unroll
                for (int b = 0; b < P; ++b) {
                    PairOut o = pair_eval<ZCLASS>(m, t[a], z[a], pb[b], pz[b]);
                    bool v = ok[a] && pok[b];
                    if (v) {
                        F[a] += o.w;
                        if (!last) T[b] -= o.w;
                        if (ITH) {
                            double cB = o.isshort ? m.k2sq : m.b_long;
                            double kk = cB * fma(o.q, o.q, 1.0);
                            Kown[a] += kk;
                            if (!last) KT[b] += kk;
                        }
                        if (count_pair) {
                            if (o.isshort) { Qs = fma(o.q, o.q, Qs); ++nshort; }
                            else           { Ql = fma(o.q, o.q, Ql); }
                            ++npair;
                            if (WF) {
                                if (o.isshort) prod *= o.fac;
                                else prodL *= o.fac;
                            }
                        }
                    }
                }
            }
            if (!last) {
                int src = (threadIdx.x & 63) - gl + ((gl + G - 1) & (G - 1));
#pragma unroll
                for (int b = 0; b < P; ++b) {
                    T[b] = __shfl(T[b], src, 64);
                    if (ITH) KT[b] = __shfl(KT[b], src, 64);
                }
            }
            // fold the running products into logs often enough that they
            // can neither underflow nor overflow
            if (WF && ((k & 3) == 0 || P > 2 || last)) {
                lsum += log(prod) + m.beta * log(prodL);
                prod = 1.0; prodL = 1.0;
            }
        }
        if (WF) lsum += (double)nshort * m.log_am;
        // deliver the travelling sums to their owners (lane gl ^ G/2 holds them)
#pragma unroll
        for (int b = 0; b < P; ++b) {
            F[b] += __shfl_xor(T[b], G / 2, 64);
            if (ITH) Kown[b] += __shfl_xor(KT[b], G / 2, 64);
        }
    }

    // ---- local energy ----
    double e_lane = 0.0;
    if (ITH) {
#pragma unroll
        for (int a = 0; a < P; ++a) {
            double e = ok[a] ? (Kown[a] + kin1[a] - F[a] * F[a]) : 0.0;
            eith[a] = e;
            e_lane += e;
        }
    } else {
        int nlong = npair - nshort;
        e_lane = 2.0 * (m.k2sq * ((double)nshort + Qs) +
                        m.b_long * ((double)nlong + Ql));
#pragma unroll
        for (int a = 0; a < P; ++a)
            if (ok[a]) e_lane += kin1[a] - F[a] * F[a];
    }
    E = group_sum<G>(e_lane);
    if (WF) logwf = group_sum<G>(lsum);
}
