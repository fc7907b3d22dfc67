/*
 * qmcwalk.h -- C-ABI of libqmcwalk.so, the MI355X (gfx950) walker-propagation
 * engine for the PhD-QMCLib `mrbp_qmc` VMC/DMC sampling hot path.
 *
 * The reference has no FFI: its boundary is the Python object protocol
 * `Sampling.core_funcs` / `Sampling.blocks()` (numba-compiled callables,
 * qmc_base/vmc.py:204-257, qmc_base/dmc.py:297-370).  Each entry point below
 * names the reference interface it stands in for (paths under
 * /root/reference/src/phd_qmclib/).  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error;
 *     `qmc_last_error()` gives the message (thread-local).
 *   - "host" pointers are caller-owned CPU buffers; "dev" pointers are device
 *     (HBM) addresses.  All device memory of an ensemble is owned by its handle.
 *   - a handle is bound to one HIP device and one stream and is not
 *     thread-safe.  `stream` is a hipStream_t passed as void*.
 *     qmc_engine_create: NULL = the engine creates its own non-blocking
 *     stream; qmc_engine_create_on_stream: the caller's stream as it is, NULL
 *     being the legacy default stream (what torch.cuda.current_stream() is
 *     unless a side stream is current) -- use it whenever another library
 *     (torch.distributed / RCCL) must be stream-ordered with the engine.
 *   - configurations use the reference's row layout: positions pos[W][N],
 *     full system configurations confs[W][2][N] (row 0 position, row 1 drift;
 *     qmc_base/jastrow/model.py:31-38).
 */
#ifndef QMCWALK_H
#define QMCWALK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QMCWALK_ABI_VERSION 1

typedef struct qmc_engine qmc_engine;
typedef struct qmc_vmc qmc_vmc;
typedef struct qmc_dmc qmc_dmc;

/* mrbp_qmc/model.py:40-75 -- Params, OBFParams, TBFParams flattened in
 * declaration order (what `Spec.cfc_spec` hands to every core function). */
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
} qmc_model_params;

/* mrbp_qmc/vmc.py:29-38 (TPFParams) + Sampling fields :70-80; `gaussian`
 * selects the vmc_ndf proposal (mrbp_qmc/vmc_ndf.py:23-51, move_spread =
 * sigma = sqrt(time_step)). */
typedef struct {
    int64_t num_chains;
    double move_spread;
    uint64_t rng_seed;
    uint32_t chain0;      /* Philox slot of chain 0 (rank offset)          */
    int32_t gaussian;
} qmc_vmc_params;

/* mrbp_qmc/dmc.py:143-185 (Sampling fields + DDFParams) */
typedef struct {
    int64_t max_num_walkers;
    int64_t target_num_walkers;
    double time_step;
    double num_walkers_control_factor;
    uint64_t rng_seed;
    uint32_t slot0;            /* Philox slot of walker slot 0 (rank offset) */
    int32_t fix_stale_energy;  /* 0: reference behaviour         */
    int32_t external_reduce;   /* 1: E_t/W_t are reduced across ranks by the
                                  caller between step_local and step_finish  */
    int32_t reserved;
} qmc_dmc_params;

/* Estimator specs of a DMC sampling (mrbp_qmc/dmc.py:103-140, 187-225):
 * static structure factor over num_modes momenta k_m = 2 pi m / L and density
 * histogram over num_bins bins, each mixed or pure (forward walking over
 * pfw_num_time_steps).  0 modes / bins disables an estimator. */
typedef struct {
    int32_t num_modes;
    int32_t ssf_pure;
    int64_t ssf_pfw;
    int32_t num_bins;
    int32_t dens_pure;
    int64_t dens_pfw;
} qmc_dmc_est_params;

const char *qmc_last_error(void);
int qmc_abi_version(void);
/* Identity of the kernels this library was built from: the first 16 hex digits
 * of sha256 over csrc/ sources + compiler flags (csrc/Makefile).  Measurement
 * records (profiles/traffic.json) carry it, so that a figure measured on other
 * kernels is recognised as stale. */
const char *qmc_source_hash(void);
int qmc_device_count(int *count);

/* Diagnostic (no GPU needed): the piecewise-polynomial table of the one-body
 * factor (mrbp_qmc/model.py:404-464: f1'/f1 and log f1 inside the unit cell)
 * the kernels would use for this model: intervals over the well and over the
 * barrier (0, 0 = the closed forms are evaluated directly) and the worst
 * deviation from the closed forms found while building it. */
int qmc_model_one_body_table_info(const qmc_model_params *model,
                                  int32_t *rows_well, int32_t *rows_barrier,
                                  double *max_err);

/* Diagnostic (no GPU needed): the row table of the pair-function angles
 * (sin / cos of pi z / L and of k2 z, mrbp_qmc/model.py:466-531) the kernels
 * would use for this model: rows over [0, L) (0 = the angles are evaluated by
 * polynomial `sincos` directly) and the worst absolute deviation of the
 * kernels' row + angle-addition arithmetic from long-double sin / cos. */
int qmc_model_trig_table_info(const qmc_model_params *model, int32_t *rows,
                              double *max_err);

/* Diagnostic (no GPU needed): the kernels' natural logarithm (row table of
 * 1/c and log c + five terms of log1p; it takes the place of numpy's log in
 * log|psi| = sum log f, the Metropolis test log(u) < 2 (log|psi'| - log|psi|),
 * qmc_base/vmc.py:636, and the Box-Muller normals) restated on the host: rows
 * and the worst |error| / (1 + |log x|) against long double over 2*10^5
 * arguments of every size. */
int qmc_log_table_info(int32_t *rows, double *max_err);

/* ---- engine: model constants on one device --------------------------- */
int qmc_engine_create(const qmc_model_params *model, int device, void *stream,
                      qmc_engine **out);
int qmc_engine_create_on_stream(const qmc_model_params *model, int device,
                                void *stream, qmc_engine **out);
void qmc_engine_destroy(qmc_engine *eng);
/* the stream the engine launches on; *owned = 1 if the engine created it */
int qmc_engine_stream(qmc_engine *eng, void **stream, int *owned);
/* The reference's reduced-precision knob, `jit_fastmath`
 * (mrbp_qmc/dmc.py:159-160; qmc_base/jastrow/dmc.py `fastmath=`): with on != 0
 * the O(N^2) pair loop (products of the per-particle sin/cos tables, the
 * quotient per pair, the pair sums) runs in float; positions, the tables
 * themselves, the one-body factor, the energy assembly, the logarithms and
 * the Metropolis test stay in double.  Off by default; relative error of the
 * local energy ~1e-6 (tests/test_gpu_fastmath.py reports it).  *in_effect
 * tells whether the engine has the variant for its model (boson_number > 32,
 * cutoff not close to L/2); where it has not, the call changes nothing. */
int qmc_engine_set_fast_math(qmc_engine *eng, int on, int *in_effect);
int qmc_engine_sync(qmc_engine *eng);
/* HIP-event timing on the engine's stream (bench / roofline) */
int qmc_engine_timer_start(qmc_engine *eng);
int qmc_engine_timer_stop(qmc_engine *eng, float *elapsed_ms);
/* Kernel profile: while open, every launch of the dominant kernel of a step
 * (vmc_step_kernel / dmc_evolve_kernel) is bracketed by its own HIP event
 * pair on the engine's stream (at most max_launches of them); profile_end
 * synchronises and returns the number of launches, the sum, the shortest and
 * the longest of their durations.  bench.py derives roofline.achieved from
 * it. */
int qmc_engine_profile_begin(qmc_engine *eng, int64_t max_launches);
int qmc_engine_profile_end(qmc_engine *eng, int64_t *launches,
                           double *total_ms, double *min_ms, double *max_ms);

/* Diagnostic (no reference counterpart): a library built
 * with -DQMC_TIMING stamps the shader clock at the section marks of the walker
 * kernels; this returns, per section, the cycles of wavefront lifetime spent
 * in it and the number of visits since the last reset (nsec must be 32;
 * entries 16..31 are the sections inside the energy pass of the VMC step).
 * The shipped library carries no stamps and returns an error. */
int qmc_engine_section_profile(qmc_engine *eng, uint64_t *cycles,
                               uint64_t *visits, int32_t nsec, int32_t reset);
const char *qmc_section_name(int32_t section);
/* Diagnostic: a library built with -DQMC_CUTS ends every wavefront of the walker
 * kernels at the section mark selected here (-1: never); hardware counters of
 * runs cut at successive marks give per-section executed instructions.  The
 * shipped library returns an error. */
int qmc_engine_section_cut(qmc_engine *eng, int32_t section);
/* Diagnostic counters of the shipped kernels (test-visible evidence of which
 * code path ran; no reference counterpart).  out[0]: walkers of the
 * one-wavefront-per-walker shapes (33 <= N <= 128) that failed the per-walker
 * checks of the sorted-row pair sums (csrc/qmc_sorted64.h, qmc_sorted128.h) in
 * a VMC / DMC stepping kernel and were evaluated by the general pair sum
 * inside the same kernel, since the last reset.  n <= 4; synchronises. */
int qmc_engine_diag_counters(qmc_engine *eng, uint64_t *out, int32_t n,
                             int32_t reset);

/* Stands in for model.core_funcs.{wf_abs_log, energy, drift,
 * ith_energy_and_drift} (qmc_base/jastrow/model.py:298-366, 476-564, 756-773,
 * 793-854) over a batch of configurations.  Host buffers; any output may be
 * NULL.  pos[nconf][N] -> wf[nconf], energy[nconf], ith_energy[nconf][N],
 * drift[nconf][N]. */
int qmc_evaluate(qmc_engine *eng, int64_t nconf, const double *pos,
                 double *wf_abs_log, double *energy, double *ith_energy,
                 double *drift);
/* Same on device-resident buffers, asynchronous on the engine's stream. */
int qmc_evaluate_dev(qmc_engine *eng, int64_t nconf, const double *pos_dev,
                     double *wf_dev, double *energy_dev, double *ith_dev,
                     double *drift_dev);

/* Plain device buffers, so that a configuration set can stay resident across
 * many qmc_evaluate_dev calls with different engines: the correlated-sampling
 * optimiser re-evaluates wf_abs_log / energy of one fixed set for every trial
 * value of the variational parameter (mrbp_qmc/model.py:818-942,
 * qmc_base/jastrow/model.py:1125-1206).  upload/download are synchronous. */
int qmc_buffer_alloc(int device, size_t bytes, void **out);
int qmc_buffer_free(void *buf);
int qmc_buffer_upload(void *dst_dev, const void *src_host, size_t bytes);
int qmc_buffer_download(void *dst_host, const void *src_dev, size_t bytes);

/* ---- VMC ensemble of independent Metropolis chains -------------------- */
/* vmc.Sampling + core_funcs.states_generator/blocks (qmc_base/vmc.py:557-648,
 * 670-770; jastrow/vmc.py:169-264; mrbp_qmc/vmc.py:174-233). */
int qmc_vmc_create(qmc_engine *eng, const qmc_vmc_params *p, qmc_vmc **out);
void qmc_vmc_destroy(qmc_vmc *v);
/* build_state (mrbp_qmc/vmc.py:145-165) for every chain: uploads pos[W][N],
 * evaluates log|psi|, resets the step counter; the next block's first yield
 * is the initial state flagged ACCEPTED (qmc_base/vmc.py:616-618). */
int qmc_vmc_set_state(qmc_vmc *v, const double *pos);
int qmc_vmc_get_state(qmc_vmc *v, double *pos, double *wf_abs_log,
                      double *energy_carry);
/* One block of `nyield` generator yields per chain.  Per-chain block sums
 * (sum of energy, of energy^2, accepted count) are always produced on the
 * device; the host copies requested here synchronise.  Series buffers are
 * [nyield][W] (step-major) host arrays or NULL; series_pos is [nyield][W][N]
 * (the chain configurations `as_chain` returns, qmc_base/vmc.py:785-900). */
int qmc_vmc_run_block(qmc_vmc *v, int64_t nyield, double *sum_energy,
                      double *sum_energy2, int64_t *num_accepted,
                      double *series_wf, double *series_energy,
                      uint8_t *series_stat, double *series_pos);
/* Device addresses of the chain state, pos[W][N] and wf[W].  NOTE: on the
 * device the particles of a configuration are kept in position order (an
 * internal label array maps lanes back to the caller's particle indices;
 * every host-facing call un-permutes).  Symmetric functions of a
 * configuration can be evaluated on these buffers directly. */
int qmc_vmc_state_dev(qmc_vmc *v, double **pos, double **wf);
/* Static structure factor parts of the CURRENT configurations summed over the
 * chains: out[num_modes][3] = sum_w (|rho_m|^2, Re rho_m, Im rho_m),
 * rho_m = sum_i exp(i 2 pi m z_i / L)  (the per-step quantity of
 * qmc_base/jastrow/vmc.py:304-351, evaluated for a whole ensemble in one
 * launch).  Host buffer; synchronous. */
int qmc_vmc_ssf(qmc_vmc *v, int32_t num_modes, double *out);
/* Device addresses of the per-chain block sums of the last block
 * (sum_e[W], sum_e2[W], n_acc[W]) for on-device reductions / collectives. */
int qmc_vmc_block_sums_dev(qmc_vmc *v, double **sum_e, double **sum_e2,
                           int64_t **n_acc);
/* TEST ONLY: replay a recorded random stream instead of Philox.  tape is
 * host [W][steps][N + 1] (N proposal draws, then the accept uniform). */
int qmc_vmc_set_tape(qmc_vmc *v, const double *tape, int64_t steps);

/* ---- DMC walker ensemble ---------------------------------------------- */
/* dmc.Sampling + core_funcs.states_generator/blocks (qmc_base/dmc.py:614-787,
 * 815-971; jastrow/dmc.py:634-951, 1030-1174; mrbp_qmc/dmc.py:268-328). */
int qmc_dmc_create(qmc_engine *eng, const qmc_dmc_params *p, qmc_dmc **out);
void qmc_dmc_destroy(qmc_dmc *d);
/* build_state (mrbp_qmc/dmc.py:268-328): uploads pos[nw][N], computes energy
 * and drift of every walker, unit weights; ref_energy = mean energy unless
 * use_ref_energy.  The caller has already applied `[-target_num_walkers:]`. */
int qmc_dmc_set_state(qmc_dmc *d, int64_t nw, const double *pos,
                      int use_ref_energy, double ref_energy);
/* build_state from positions already in HBM; set_state_from_vmc takes the
 * first nw chains of a VMC ensemble of the same engine (VMC -> DMC hand-off
 * without a host round trip, tests/mrbp_qmc/test_dmc.py:76-83); when nw
 * exceeds the number of chains they are reused cyclically. */
int qmc_dmc_set_state_dev(qmc_dmc *d, int64_t nw, const double *pos_dev,
                          int use_ref_energy, double ref_energy);
int qmc_dmc_set_state_from_vmc(qmc_dmc *d, qmc_vmc *v, int64_t nw,
                               int use_ref_energy, double ref_energy);
/* Restart from a yielded State (qmc_base/dmc.py:707-716): confs[nw][2][N],
 * energy[nw], weight[nw] copied as they are; slot_energy[maxw] (or NULL) is the
 * whole props.energy array of that State. */
int qmc_dmc_set_full_state(qmc_dmc *d, int64_t nw, const double *confs,
                           const double *energy, const double *weight,
                           const double *slot_energy, double ref_energy);
/* `nsteps` generator iterations (branch -> diffuse/evaluate -> estimators ->
 * E_ref feedback).  Per-step series are host arrays of length nsteps or NULL
 * (PropsData, qmc_base/dmc.py:130-143). */
int qmc_dmc_run_block(qmc_dmc *d, int64_t nsteps, double *energy,
                      double *weight, uint64_t *num_walkers,
                      double *ref_energy, double *accum_energy);
/* Estimators of core_funcs.blocks (qmc_base/dmc.py:897-940;
 * qmc_base/jastrow/dmc.py:195-302, 363-631; mrbp_qmc/dmc.py:472-547).
 * run_block_est = run_block + per-step iter_ssf[nsteps][num_modes][3] /
 * iter_density[nsteps][num_bins] (host, may be NULL); the per-walker
 * forward-walking buffers are reset at the start of every block and the
 * estimators are skipped when eval_estimators == 0 (burn-in blocks). */
int qmc_dmc_set_estimators(qmc_dmc *d, const qmc_dmc_est_params *p);
int qmc_dmc_run_block_est(qmc_dmc *d, int64_t nsteps, int eval_estimators,
                          double *energy, double *weight, uint64_t *num_walkers,
                          double *ref_energy, double *accum_energy,
                          double *iter_ssf, double *iter_density);
/* The same for split-step (multi-GPU) drivers: est_begin_block opens a block
 * of nsteps steps (per-block resets), step_estimators evaluates the
 * estimators on the population yielded by the last step_finish into row
 * step_idx of this rank's iter buffers, est_iter_dev returns their device
 * addresses (iter_ssf[nsteps][num_modes][3], iter_density[nsteps][num_bins];
 * sums over this rank's walkers: linear, so ranks add them up). */
int qmc_dmc_est_begin_block(qmc_dmc *d, int64_t nsteps);
int qmc_dmc_step_estimators(qmc_dmc *d, int64_t step_idx);
int qmc_dmc_est_iter_dev(qmc_dmc *d, double **iter_ssf, double **iter_density);
/* The yielded ("actual") State after the last step (qmc_base/dmc.py:773-780):
 * confs[maxw][2][N], energy/weight[maxw], mask[maxw], cloning_ref[maxw];
 * scalars[5] = energy, weight, ref_energy, accum_energy, num_walkers. */
int qmc_dmc_get_state(qmc_dmc *d, double *confs, double *energy,
                      double *weight, uint8_t *mask, int64_t *cloning_ref,
                      double *scalars);
/* Split step for multi-GPU runs (external_reduce = 1): step_local runs the
 * branching and propagation of this rank's walkers and leaves this rank's
 * (E_t, W_t) in partial_dev[0..1]; the caller all-reduces them (RCCL) into
 * total_dev[0..1] on the same stream; step_finish applies the E_ref feedback
 * with the global sums and the global target. */
int qmc_dmc_step_local(qmc_dmc *d, double *partial_dev);
int qmc_dmc_step_finish(qmc_dmc *d, const double *total_dev);
int qmc_dmc_read_series(qmc_dmc *d, int64_t nsteps, double *energy,
                        double *weight, uint64_t *num_walkers,
                        double *ref_energy, double *accum_energy);
/* Population rebalance.  A walker record is walker_record_size doubles: pos[N],
 * drift[N], lane labels[N], energy, LOG of the branching weight (the device's
 * own storage form: records are opaque between engines of one build) and,
 * when estimators are set, the
 * walker's forward-walking rows (S(k) parts [num_modes][3], density
 * [num_bins]).  export packs walkers [first, first+count) of the current
 * population into buf_dev; import_walkers_at writes `count` records into
 * slots [first, first+count) and makes first+count the population size;
 * set_num_walkers drops the tail.  These three are stream-ordered (no host
 * synchronisation: the caller planned the transfer from num_walkers of every
 * rank).  num_walkers, import_walkers (append at the device's count) and
 * truncate (checked against it) read the device and synchronise. */
int qmc_dmc_num_walkers(qmc_dmc *d, int64_t *nw);
int qmc_dmc_walker_record_size(qmc_dmc *d, int64_t *doubles);
int qmc_dmc_export_walkers(qmc_dmc *d, int64_t first, int64_t count,
                           double *buf_dev);
int qmc_dmc_import_walkers_at(qmc_dmc *d, int64_t first, int64_t count,
                              const double *buf_dev);
int qmc_dmc_set_num_walkers(qmc_dmc *d, int64_t nw);
int qmc_dmc_import_walkers(qmc_dmc *d, int64_t count, const double *buf_dev);
int qmc_dmc_truncate(qmc_dmc *d, int64_t new_nw);
/* TEST ONLY: replay recorded streams.  u[] / g[] are host arrays; step t
 * reads its branching uniforms at u[u_off[t] + parent] and its standard
 * normals at g[g_off[t] + slot*N + i]. */
int qmc_dmc_set_tape(qmc_dmc *d, const double *u, int64_t nu, const double *g,
                     int64_t ng, const int64_t *u_off, const int64_t *g_off,
                     int64_t nsteps);

#ifdef __cplusplus
}
#endif
#endif
