/*
 * qmc_oracle.h -- CPU restatement of the PhD-QMCLib mrbp_qmc VMC/DMC sampling
 * hot path.  TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product package never
 * does (it fails loudly when the HIP library is missing).
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit (or to
 * the stated tolerance) against tests/golden/ *.npz, which were produced by
 * running the reference's own function bodies (oracle/refgen/gen_golden.py).
 *
 * Citations are paths under /root/reference/src/phd_qmclib/.
 */
#ifndef QMC_ORACLE_H
#define QMC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mrbp_qmc/model.py:40-75 (Params + OBFParams + TBFParams, flattened) */
typedef struct {
    double lattice_depth;
    double lattice_ratio;
    double interaction_strength;
    int64_t boson_number;
    double supercell_size;
    double tbf_contact_cutoff;
    double defect_magnitude;
    int64_t defects_sep;
    double well_width;
    double barrier_width;
    int64_t is_free;
    int64_t is_ideal;
    double param_e0;
    double param_k1;
    double param_kp1;
    double param_k2;
    double param_beta;
    double param_r_off;
    double param_am;
} orc_model;

/* Counter RNG shared (as an algorithm) with the device engine: Philox4x32-10,
 * key = seed, counter = (slot, step, index, stream) -- DMC branching, the
 * Gaussian VMC proposal -- and Philox2x32-10 for the uniform VMC proposal and
 * the DMC diffusion normals (below). */
enum { ORC_STREAM_VMC_MOVE = 0, ORC_STREAM_VMC_ACCEPT = 1,
       ORC_STREAM_DMC_BRANCH = 2, ORC_STREAM_DMC_DIFFUSE = 3 };
void orc_philox_uniform2(uint64_t seed, uint32_t slot, uint32_t step,
                         uint32_t index, uint32_t stream, double *u);
void orc_philox_normal2(uint64_t seed, uint32_t slot, uint32_t step,
                        uint32_t index, uint32_t stream, double *g);
/* Philox2x32-10, one 64-bit block: the uniform proposal of the VMC step.
 * orc_vmc_move_block: the block of (seed; chain slot, step, particle);
 * word 0 -> displacement / move_spread = (w0 + 1/2) 2^-32 - 1/2; the accept
 * draw of a step = u53(word 1 of particle 0, word 1 of particle 1). */
void orc_philox2x32(uint32_t c0, uint32_t c1, uint32_t key, uint32_t *out);
void orc_vmc_move_block(uint64_t seed, uint32_t slot, uint32_t step,
                        uint32_t index, uint32_t *w);
double orc_vmc_move_unit(uint32_t w0);
double orc_vmc_accept_uniform(uint32_t w1_p0, uint32_t w1_p1);
/* DMC diffusion normal: steps 2m / 2m+1 share one block (cos / sin branch) */
void orc_dmc_normal2(uint64_t seed, uint32_t slot, uint32_t step2,
                     uint32_t index, double *g);
double orc_dmc_normal(uint64_t seed, uint32_t slot, uint32_t step,
                      uint32_t index);
double orc_philox_normal(uint64_t seed, uint32_t slot, uint32_t step,
                         uint32_t index, uint32_t stream);

/* qmc_base/jastrow/model.py:298-366 */
double orc_wf_abs_log(const orc_model *m, const double *pos);
/* qmc_base/jastrow/model.py:793-854 for every i; returns sum_i e_i
 * (== `energy`, :756-773).  ith_energy / drift may be NULL. */
double orc_energy_drift(const orc_model *m, const double *pos,
                        double *ith_energy, double *drift);
/* numpy's pairwise float64 add.reduce (what `arr[:n].sum()` does) */
double orc_np_sum(const double *a, int64_t n);

/* ---- VMC (qmc_base/vmc.py:571-646, 686-768; jastrow/vmc.py:208-262) ---- */
typedef struct {
    double move_spread;
    uint64_t seed;
    uint32_t chain;       /* Philox slot of this chain                    */
    uint32_t step0;       /* global index of the first real step          */
    int32_t yield_initial;/* first yield is the initial state (ACCEPTED)  */
    int32_t gaussian;     /* vmc_ndf proposal: normal(0, sqrt(time_step)) */
} orc_vmc_cfg;

/* Runs `nyield` yields of one chain.  pos[N], *wf, *e_prev are updated in
 * place (e_prev = energy carried to a rejected move).  tape: N+1 uniforms per
 * real step (N proposals then the accept draw) or NULL for Philox.  Any of
 * the out_* series may be NULL.  Returns the number of ACCEPTED yields. */
int64_t orc_vmc_chain(const orc_model *m, const orc_vmc_cfg *cfg,
                      double *pos, double *wf, double *e_prev,
                      int64_t nyield, const double *tape,
                      double *out_wf, double *out_energy, uint8_t *out_stat);

/* Ensemble of independent chains (OpenMP over chains): per-chain sums over
 * the nyield yields.  pos is [nchains][N]. */
void orc_vmc_ensemble(const orc_model *m, const orc_vmc_cfg *cfg,
                      int64_t nchains, double *pos, double *wf,
                      double *e_prev, int64_t nyield,
                      double *sum_e, double *sum_e2, int64_t *n_acc,
                      int nthreads);

/* ---- DMC (qmc_base/dmc.py:622-653, 679-785; jastrow/dmc.py:645-671,
 *           758-825, 847-949) ---- */
typedef struct {
    int64_t max_num_walkers;
    int64_t target_num_walkers;
    double time_step;
    double control_factor;   /* num_walkers_control_factor (kappa)        */
    uint64_t seed;
    uint32_t slot0;          /* Philox slot offset (global slot of slot 0)*/
    int32_t fix_stale_energy;/* 0 = reference quirk D1, 1 = parent energy */
} orc_dmc_cfg;

typedef struct {
    /* three StateData buffers, confs are [maxw][2][N] like the reference */
    double *prev_confs, *prev_energy, *prev_weight;
    double *actual_confs, *actual_energy, *actual_weight;
    double *next_confs, *next_energy, *next_weight;
    uint8_t *actual_mask;
    int64_t *cloning_ref;
    int64_t prev_num_walkers;
    double ref_energy;
    double total_energy, total_weight;
    uint32_t step;           /* global step counter (Philox)              */
} orc_dmc_state;

typedef struct {
    double energy, weight;
    int64_t num_walkers;
    double ref_energy, accum_energy;
    int64_t n_uniform, n_normal;   /* draws consumed by this step        */
} orc_dmc_yield;

/* One generator iteration.  u_tape / g_tape may be NULL (Philox).  After the
 * call `actual_*` hold the yielded state and prev/next have been exchanged. */
void orc_dmc_step(const orc_model *m, const orc_dmc_cfg *cfg,
                  orc_dmc_state *st, const double *u_tape,
                  const double *g_tape, orc_dmc_yield *out, int nthreads);

/* qmc_base/jastrow/dmc.py:1043-1078 for walkers [0, n): energy + drift of
 * the initial configurations; confs is [n][2][N] (pos row in, drift row out) */
void orc_dmc_prepare(const orc_model *m, int64_t n, double *confs,
                     double *energy, int nthreads);

int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
