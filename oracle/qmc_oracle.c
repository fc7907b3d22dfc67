/*
 * qmc_oracle.c -- CPU restatement of the PhD-QMCLib mrbp_qmc VMC/DMC sampling
 * hot path, in the reference's own operation order (full N(N-1) ordered pair
 * loops, libm tan/pow/log/exp, Python floor-mod, serial branching table).
 *
 * TEST INFRASTRUCTURE ONLY -- see qmc_oracle.h.  Parity status: PINNED against
 * tests/golden/ (vectors produced by the reference's own function bodies).
 *
 * Citations are paths under /root/reference/src/phd_qmclib/.
 * Build: see oracle/Makefile (gcc -O2 -fopenmp, no -ffast-math: the reference
 * default is jit_fastmath=False, mrbp_qmc/dmc.py:160).
 */
#include "qmc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* Python float semantics                                              */
/* ------------------------------------------------------------------ */

/* float.__mod__ (floor-mod): sign of the result follows the divisor.
 * Used by qmc_base/utils.py:50,66 and mrbp_qmc/model.py:417,440,462. */
static double py_mod(double a, double b)
{
    double m = fmod(a, b);
    if (m != 0.0) {
        if ((b < 0.0) != (m < 0.0))
            m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}

/* divmod(a, b) for floats (mrbp_qmc/model.py:546) */
static void py_divmod(double a, double b, double *q, double *r)
{
    double m = fmod(a, b);
    double div = (a - m) / b;
    if (m != 0.0) {
        if ((b < 0.0) != (m < 0.0)) {
            m += b;
            div -= 1.0;
        }
    } else {
        m = copysign(0.0, b);
    }
    double fl;
    if (div != 0.0) {
        fl = floor(div);
        if (div - fl > 0.5)
            fl += 1.0;
    } else {
        fl = copysign(0.0, a / b);
    }
    *q = fl;
    *r = m;
}

/* x ** 2 on a Python float is libm pow(x, 2.0) */
static inline double py_sqr(double x) { return pow(x, 2.0); }

/* ------------------------------------------------------------------ */
/* Philox4x32-10 counter RNG (Salmon et al., SC'11 -- public algorithm) */
/* ------------------------------------------------------------------ */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c[0];
        uint64_t p1 = (uint64_t)M1 * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += W0; k1 += W1;
    }
}

static inline double u53(uint32_t hi, uint32_t lo)
{
    /* 53 random bits -> [0, 1) */
    uint64_t b = ((uint64_t)(hi >> 5) << 26) | (uint64_t)(lo >> 6);
    return (double)b * (1.0 / 9007199254740992.0);
}

void orc_philox_uniform2(uint64_t seed, uint32_t slot, uint32_t step,
                         uint32_t index, uint32_t stream, double *u)
{
    uint32_t c[4] = { slot, step, index, stream };
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    u[0] = u53(c[0], c[1]);
    u[1] = u53(c[2], c[3]);
}

/* Philox2x32-10 (same paper): one 64-bit block per call.  The uniform
 * proposal of the VMC step draws from it -- a particle needs ONE number per
 * step, and the device pays per instruction issued, not per lane served: the
 * 4x32 generator computed 128 bits per lane to use 64 of them. */
static void philox2x32_10(uint32_t c[2], uint32_t k)
{
    const uint32_t M = 0xD256D193u, W = 0x9E3779B9u;
    for (int r = 0; r < 10; ++r) {
        uint64_t p = (uint64_t)M * c[0];
        uint32_t n0 = (uint32_t)(p >> 32) ^ k ^ c[1];
        c[1] = (uint32_t)p;
        c[0] = n0;
        k += W;
    }
}

void orc_philox2x32(uint32_t c0, uint32_t c1, uint32_t key, uint32_t *out)
{
    uint32_t c[2] = { c0, c1 };
    philox2x32_10(c, key);
    out[0] = c[0];
    out[1] = c[1];
}

/* The VMC move stream: (key; counter) of the block of particle `index` of
 * chain `slot` at Metropolis step `step`.  Every (slot < 2^28, step < 2^26,
 * index < 1024) has its own counter under the key of the seed; bits beyond
 * those ranges move into the key (distinct blocks again, as far as 32 bits of
 * key allow).  The 64-bit seed is folded into the 32-bit key. */
void orc_vmc_move_block(uint64_t seed, uint32_t slot, uint32_t step,
                        uint32_t index, uint32_t *w)
{
    uint32_t key = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x85EBCA6Bu);
    key += (step >> 26) * 0x632BE5ABu + (slot >> 28) * 0xC2B2AE35u;
    uint32_t c[2];
    c[0] = ((step & 0x3FFFFFFu) << 6) | ((slot >> 22) & 0x3Fu);
    c[1] = ((slot & 0x3FFFFFu) << 10) | (index & 0x3FFu);
    philox2x32_10(c, key);
    w[0] = c[0];
    w[1] = c[1];
}

/* word 0 -> the particle's displacement in units of the move spread,
 * (w0 + 1/2) 2^-32 - 1/2 in (-1/2, 1/2): 32 random bits, every step exact */
double orc_vmc_move_unit(uint32_t w0)
{
    return ((double)w0 + 0.5) * (1.0 / 4294967296.0) - 0.5;
}

/* the accept draw of a step: 53 bits from the second words of the blocks of
 * particles 0 and 1 (of particle 0 twice in a one-particle model) */
double orc_vmc_accept_uniform(uint32_t w1_p0, uint32_t w1_p1)
{
    return u53(w1_p0, w1_p1);
}

void orc_philox_normal2(uint64_t seed, uint32_t slot, uint32_t step,
                        uint32_t index, uint32_t stream, double *g)
{
    /* Box-Muller pair; 1-u keeps the log argument in (0, 1] */
    double u[2];
    orc_philox_uniform2(seed, slot, step, index, stream, u);
    double r = sqrt(-2.0 * log(1.0 - u[0]));
    g[0] = r * cos(6.283185307179586476925 * u[1]);
    g[1] = r * sin(6.283185307179586476925 * u[1]);
}

double orc_philox_normal(uint64_t seed, uint32_t slot, uint32_t step,
                         uint32_t index, uint32_t stream)
{
    /* DMC diffusion keying: time steps 2m and 2m+1 share one Philox block,
     * cosine branch on even steps, sine branch on odd ones */
    double g[2];
    orc_philox_normal2(seed, slot, step >> 1, index, stream, g);
    return g[step & 1u];
}

/* The DMC diffusion stream: one Philox2x32-10 block per (walker slot, pair of
 * time steps, particle) -- the counter packing of the VMC move blocks under
 * another key -- and both Box-Muller normals of it: the cosine branch at time
 * step 2m, the sine branch at 2m + 1.  32-bit uniforms (w + 1/2) 2^-32. */
void orc_dmc_normal2(uint64_t seed, uint32_t slot, uint32_t step2,
                     uint32_t index, double *g)
{
    uint32_t key = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x85EBCA6Bu);
    key += 0x27D4EB2Fu;
    key += (step2 >> 26) * 0x632BE5ABu + (slot >> 28) * 0xC2B2AE35u;
    uint32_t c[2];
    c[0] = ((step2 & 0x3FFFFFFu) << 6) | ((slot >> 22) & 0x3Fu);
    c[1] = ((slot & 0x3FFFFFu) << 10) | (index & 0x3FFu);
    philox2x32_10(c, key);
    double u0 = ((double)c[0] + 0.5) * (1.0 / 4294967296.0);
    double u1 = ((double)c[1] + 0.5) * (1.0 / 4294967296.0);
    double r = sqrt(-2.0 * log(u0));
    g[0] = r * cos(6.283185307179586476925 * u1);
    g[1] = r * sin(6.283185307179586476925 * u1);
}

double orc_dmc_normal(uint64_t seed, uint32_t slot, uint32_t step,
                      uint32_t index)
{
    double g[2];
    orc_dmc_normal2(seed, slot, step >> 1, index, g);
    return g[step & 1u];
}

/* ------------------------------------------------------------------ */
/* Model functions                                                     */
/* ------------------------------------------------------------------ */

/* qmc_base/utils.py:35-51 */
static double min_distance(double z_i, double z_j, double sc_size)
{
    double sc_half = 0.5 * sc_size;
    double z_ij = z_i - z_j;
    if (fabs(z_ij) > sc_half)
        return -sc_half + py_mod(z_ij + sc_half, sc_size);
    return z_ij;
}

/* qmc_base/utils.py:55-66 with (z_min, z_max) = (0, L) */
static double recast(double z, double z_min, double z_max)
{
    double sc_size = z_max - z_min;
    return z_min + py_mod(z - z_min, sc_size);
}

/* mrbp_qmc/model.py:404-425 */
static double one_body_func(double z, const orc_model *m)
{
    double v0 = m->lattice_depth, r = m->lattice_ratio;
    double e0 = m->param_e0, k1 = m->param_k1, kp1 = m->param_kp1;
    double z_cell = py_mod(z, 1.0);
    double z_a = 1 / (1 + r), z_b = r / (1 + r);
    if (z_a < z_cell)
        return cosh(kp1 * (z_cell - 1. + 0.5 * z_b));
    double cf = sqrt(1 + v0 / e0 * pow(sinh(0.5 * sqrt(v0 - e0) * z_b), 2.0));
    return cf * cos(k1 * (z_cell - 0.5 * z_a));
}

/* mrbp_qmc/model.py:428-447 */
static double one_body_log_dz(double z, const orc_model *m)
{
    double r = m->lattice_ratio, k1 = m->param_k1, kp1 = m->param_kp1;
    double z_cell = py_mod(z, 1.0);
    double z_a = 1 / (1 + r), z_b = r / (1 + r);
    if (z_a < z_cell)
        return kp1 * tanh(kp1 * (z_cell - 1. + 0.5 * z_b));
    return -k1 * tan(k1 * (z_cell - 0.5 * z_a));
}

/* mrbp_qmc/model.py:450-464 */
static double one_body_log_dz2(double z, const orc_model *m)
{
    double v0 = m->lattice_depth, r = m->lattice_ratio, e0 = m->param_e0;
    double z_cell = py_mod(z, 1.0);
    double z_a = 1 / (1 + r);
    return z_a < z_cell ? v0 - e0 : -e0;
}

/* mrbp_qmc/model.py:467-486 */
static double two_body_func(double rz, const orc_model *m)
{
    double L = m->supercell_size, rm = m->tbf_contact_cutoff;
    if (rz < fabs(rm))
        return m->param_am * cos(m->param_k2 * (rz - m->param_r_off));
    return pow(sin(M_PI * rz / L), m->param_beta);
}

/* mrbp_qmc/model.py:489-507 */
static double two_body_log_dz(double rz, const orc_model *m)
{
    double L = m->supercell_size, rm = m->tbf_contact_cutoff;
    double k2 = m->param_k2;
    if (rz < fabs(rm))
        return -k2 * tan(k2 * (rz - m->param_r_off));
    return (M_PI / L) * m->param_beta / (tan(M_PI * rz / L));
}

/* mrbp_qmc/model.py:510-529 */
static double two_body_log_dz2(double rz, const orc_model *m)
{
    double L = m->supercell_size, rm = m->tbf_contact_cutoff;
    double k2 = m->param_k2, beta = m->param_beta;
    if (rz < fabs(rm))
        return -k2 * k2;
    return py_sqr(M_PI / L) * beta *
           ((beta - 1) / py_sqr(tan(M_PI * rz / L)) - 1);
}

/* mrbp_qmc/model.py:532-551 */
static double potential(double z, const orc_model *m)
{
    double n_cell, z_cell;
    py_divmod(z, 1.0, &n_cell, &z_cell);
    if (py_mod(n_cell, (double)m->defects_sep) == 0.0)
        return m->well_width < z_cell ? m->defect_magnitude : 0.;
    return m->well_width < z_cell ? m->lattice_depth : 0.;
}

/* qmc_base/jastrow/model.py:298-366 */
double orc_wf_abs_log(const orc_model *m, const double *pos)
{
    double wf = 0.;
    if (m->is_free && m->is_ideal)
        return wf;
    int64_t nop = m->boson_number;
    for (int64_t i = 0; i < nop; ++i) {
        double ith = 0.;
        double z_i = pos[i];
        if (!m->is_free)
            ith += log(fabs(one_body_func(z_i, m)));
        if (!m->is_ideal) {
            for (int64_t j = i + 1; j < nop; ++j) {
                double z_ij = min_distance(z_i, pos[j], m->supercell_size);
                ith += log(fabs(two_body_func(fabs(z_ij), m)));
            }
        }
        wf += ith;
    }
    return wf;
}

/* qmc_base/jastrow/model.py:793-854 */
static void ith_energy_and_drift(const orc_model *m, const double *pos,
                                 int64_t i, double *e_out, double *f_out)
{
    if (m->is_free && m->is_ideal) {
        *e_out = 0.; *f_out = 0.;
        return;
    }
    double kin = 0., pot = 0., drift = 0.;
    double z_i = pos[i];
    if (!m->is_free) {
        double ldz2 = one_body_log_dz2(z_i, m);
        double ldz = one_body_log_dz(z_i, m);
        kin += (-ldz2 + py_sqr(ldz));
        pot += potential(z_i, m);
        drift += ldz;
    }
    if (!m->is_ideal) {
        int64_t nop = m->boson_number;
        for (int64_t j = 0; j < nop; ++j) {
            if (j == i)
                continue;
            double z_ij = min_distance(z_i, pos[j], m->supercell_size);
            double sgn = copysign(1., z_ij);
            double ldz2 = two_body_log_dz2(fabs(z_ij), m);
            double ldz = two_body_log_dz(fabs(z_ij), m) * sgn;
            kin += (-ldz2 + py_sqr(ldz));
            drift += ldz;
        }
    }
    *e_out = kin - py_sqr(drift) + pot;
    *f_out = drift;
}

double orc_energy_drift(const orc_model *m, const double *pos,
                        double *ith_energy, double *drift)
{
    double energy = 0.;
    for (int64_t i = 0; i < m->boson_number; ++i) {
        double e, f;
        ith_energy_and_drift(m, pos, i, &e, &f);
        if (ith_energy) ith_energy[i] = e;
        if (drift) drift[i] = f;
        energy += e;
    }
    return energy;
}

/* numpy float64 add.reduce over a contiguous 1-D array: pairwise summation
 * in blocks of 128 with an 8-way unrolled inner loop. */
static double np_pairwise(const double *a, int64_t n)
{
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; ++i)
            res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; ++k)
            r[k] = a[k];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k)
                r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) +
                     ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i)
            res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
}

double orc_np_sum(const double *a, int64_t n)
{
    /* add.reduce starts from the identity 0. and adds the pairwise sum of the
     * whole contiguous run (checked against numpy 1.26 and 2.2) */
    if (n <= 0)
        return 0.;
    return 0. + np_pairwise(a, n);
}

/* ------------------------------------------------------------------ */
/* VMC                                                                 */
/* ------------------------------------------------------------------ */

int64_t orc_vmc_chain(const orc_model *m, const orc_vmc_cfg *cfg,
                      double *pos, double *wf, double *e_prev,
                      int64_t nyield, const double *tape,
                      double *out_wf, double *out_energy, uint8_t *out_stat)
{
    int64_t nop = m->boson_number;
    double L = m->supercell_size;
    double *prop = (double *)malloc(sizeof(double) * (size_t)nop);
    int64_t accepted = 0;
    uint32_t step = cfg->step0;
    double wf_actual = *wf;
    double e_carry = *e_prev;

    for (int64_t y = 0; y < nyield; ++y) {
        int stat;
        if (y == 0 && cfg->yield_initial) {
            /* qmc_base/vmc.py:616-618: initial state, flagged ACCEPTED */
            stat = 1;
        } else {
            /* qmc_base/jastrow/vmc.py:208-224 + mrbp_qmc/vmc.py:215-233 */
            uint32_t a_hi = 0, a_lo = 0;   /* words of the accept draw */
            for (int64_t i = 0; i < nop; ++i) {
                double d;
                if (tape) {
                    d = cfg->gaussian ? 0 + cfg->move_spread * (*tape++)
                                      : ((*tape++) - 0.5) * cfg->move_spread;
                } else if (cfg->gaussian) {
                    double g[2];
                    orc_philox_normal2(cfg->seed, cfg->chain, step,
                                       (uint32_t)i, ORC_STREAM_VMC_MOVE, g);
                    d = 0 + cfg->move_spread * g[0];
                } else {
                    uint32_t w[2];
                    orc_vmc_move_block(cfg->seed, cfg->chain, step,
                                       (uint32_t)i, w);
                    d = orc_vmc_move_unit(w[0]) * cfg->move_spread;
                    /* the accept draw: second words of particles 0 and 1 */
                    if (i == 0) a_hi = a_lo = w[1];
                    if (i == 1) a_lo = w[1];
                }
                prop[i] = recast(pos[i] + d, 0., 1. * L);
            }
            double wf_next = orc_wf_abs_log(m, prop);
            double ua;
            if (tape) {
                ua = *tape++;
            } else if (cfg->gaussian) {
                double u[2];
                orc_philox_uniform2(cfg->seed, cfg->chain, step, 0,
                                    ORC_STREAM_VMC_ACCEPT, u);
                ua = u[0];
            } else {
                ua = orc_vmc_accept_uniform(a_hi, a_lo);
            }
            stat = 0;
            /* qmc_base/vmc.py:636 */
            if (wf_next > 0.5 * log(ua) + wf_actual) {
                memcpy(pos, prop, sizeof(double) * (size_t)nop);
                wf_actual = wf_next;
                stat = 1;
            }
            ++step;
        }
        accepted += stat;
        /* qmc_base/jastrow/vmc.py:237-262 */
        double e = stat ? orc_energy_drift(m, pos, NULL, NULL) : e_carry;
        e_carry = e;
        if (out_wf) out_wf[y] = wf_actual;
        if (out_energy) out_energy[y] = e;
        if (out_stat) out_stat[y] = (uint8_t)stat;
    }
    *wf = wf_actual;
    *e_prev = e_carry;
    free(prop);
    return accepted;
}

void orc_vmc_ensemble(const orc_model *m, const orc_vmc_cfg *cfg,
                      int64_t nchains, double *pos, double *wf,
                      double *e_prev, int64_t nyield,
                      double *sum_e, double *sum_e2, int64_t *n_acc,
                      int nthreads)
{
    int64_t nop = m->boson_number;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
    for (int64_t c = 0; c < nchains; ++c) {
        orc_vmc_cfg cc = *cfg;
        cc.chain = cfg->chain + (uint32_t)c;
        double *e = (double *)malloc(sizeof(double) * (size_t)nyield);
        int64_t acc = orc_vmc_chain(m, &cc, pos + c * nop, wf + c,
                                    e_prev + c, nyield, NULL, NULL, e, NULL);
        double s = 0., s2 = 0.;
        for (int64_t y = 0; y < nyield; ++y) {
            s += e[y];
            s2 += e[y] * e[y];
        }
        sum_e[c] = s;
        sum_e2[c] = s2;
        n_acc[c] = acc;
        free(e);
    }
}

/* ------------------------------------------------------------------ */
/* DMC                                                                 */
/* ------------------------------------------------------------------ */

void orc_dmc_prepare(const orc_model *m, int64_t n, double *confs,
                     double *energy, int nthreads)
{
    int64_t nop = m->boson_number;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t s = 0; s < n; ++s) {
        double *c = confs + s * 2 * nop;
        energy[s] = orc_energy_drift(m, c, NULL, c + nop);
    }
}

void orc_dmc_step(const orc_model *m, const orc_dmc_cfg *cfg,
                  orc_dmc_state *st, const double *u_tape,
                  const double *g_tape, orc_dmc_yield *out, int nthreads)
{
    const int64_t nop = m->boson_number;
    const int64_t maxw = cfg->max_num_walkers;
    const double dt = cfg->time_step;
    const double sigma = sqrt(2 * dt);          /* mrbp_qmc/dmc.py:178 */
    const double L = m->supercell_size;
    if (nthreads < 1) nthreads = 1;

    /* --- sync_branching_spec (qmc_base/dmc.py:622-653): serial --- */
    int64_t nw = 0, n_uniform = 0;
    for (int64_t s = 0; s < st->prev_num_walkers; ++s) {
        if (nw >= maxw)
            break;
        double u;
        if (u_tape) {
            u = u_tape[n_uniform];
        } else {
            double uu[2];
            orc_philox_uniform2(cfg->seed, cfg->slot0 + (uint32_t)s, st->step,
                                0, ORC_STREAM_DMC_BRANCH, uu);
            u = uu[0];
        }
        ++n_uniform;
        int64_t clone = (int64_t)(st->prev_weight[s] + u);
        if (!clone)
            continue;
        int64_t a = nw;
        nw = nw + clone < maxw ? nw + clone : maxw;
        for (int64_t k = a; k < nw; ++k)
            st->cloning_ref[k] = s;
    }

    /* --- evolve_state_inner (qmc_base/jastrow/dmc.py:847-949) --- */
    const double ref_energy = st->ref_energy;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t s = 0; s < maxw; ++s) {
        if (s >= nw) {
            st->actual_mask[s] = 1;
            continue;
        }
        int64_t p = st->cloning_ref[s];
        const double *pc = st->prev_confs + p * 2 * nop;
        double *ac = st->actual_confs + s * 2 * nop;
        double *nc = st->next_confs + s * 2 * nop;
        double sys_energy = st->prev_energy[p];

        /* evolve_system (:758-825); ith_diffusion (:645-671) */
        for (int64_t i = 0; i < nop; ++i) {
            double g = g_tape ? g_tape[s * nop + i]
                              : orc_dmc_normal(cfg->seed,
                                               cfg->slot0 + (uint32_t)s,
                                               st->step, (uint32_t)i);
            double rnd = 0 + sigma * g;
            double z_next = pc[i] + 2 * pc[nop + i] * dt + rnd;
            double z = recast(z_next, 0., 1. * L);
            ac[i] = z;
            nc[i] = z;
        }
        /* quirk D1: the slot's own previous energy, not the parent's */
        double energy = cfg->fix_stale_energy ? sys_energy
                                              : st->actual_energy[s];
        double e_next = 0.;
        for (int64_t i = 0; i < nop; ++i) {
            double e_i, f_i;
            ith_energy_and_drift(m, ac, i, &e_i, &f_i);
            nc[nop + i] = f_i;
            e_next += e_i;
        }
        double mean_energy = (e_next + energy) / 2;
        st->next_energy[s] = e_next;
        st->next_weight[s] = exp(-dt * (mean_energy - ref_energy));

        /* cloning (:934-942) */
        memcpy(ac, pc, sizeof(double) * 2 * (size_t)nop);
        st->actual_energy[s] = sys_energy;
        st->actual_weight[s] = 1.;
        st->actual_mask[s] = 0;
    }

    /* --- estimators + E_ref feedback (qmc_base/dmc.py:759-771) --- */
    double e_t = orc_np_sum(st->actual_energy, nw);
    double w_t = orc_np_sum(st->actual_weight, nw);
    st->total_energy += e_t;
    st->total_weight += w_t;
    double accum = st->total_energy / st->total_weight;
    st->ref_energy = accum - cfg->control_factor *
                     log(w_t / (double)cfg->target_num_walkers) / dt;

    out->energy = e_t;
    out->weight = w_t;
    out->num_walkers = nw;
    out->ref_energy = st->ref_energy;
    out->accum_energy = accum;
    out->n_uniform = n_uniform;
    out->n_normal = nw * nop;

    /* exchange prev <-> next (qmc_base/dmc.py:783-785) */
    double *t;
    t = st->prev_confs;  st->prev_confs = st->next_confs;   st->next_confs = t;
    t = st->prev_energy; st->prev_energy = st->next_energy; st->next_energy = t;
    t = st->prev_weight; st->prev_weight = st->next_weight; st->next_weight = t;
    st->prev_num_walkers = nw;
    st->step += 1;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
