"""Thin object wrappers over the C-ABI handles (engine, VMC ensemble, DMC
ensemble).  numpy on the host side, HBM-resident state behind the handles.
The `Sampling` classes in `mrbp_qmc.vmc` / `mrbp_qmc.dmc` are built on these.
"""
import ctypes as C
import typing as t

import numpy as np

from . import _lib
from ._lib import (DmcEstParams, DmcParams, ModelParams, QmcError, VmcParams,
                   check, ptr,
                   _i64p, _u64p, _u8p)

__all__ = ['ModelEngine', 'VmcEnsemble', 'DmcEnsemble', 'EvalResult',
           'DeviceBuffer',
           'model_params_struct', 'QmcError']


class EvalResult(t.NamedTuple):
    wf_abs_log: np.ndarray    # [W]
    energy: np.ndarray        # [W]
    ith_energy: np.ndarray    # [W, N]
    drift: np.ndarray         # [W, N]


def model_params_struct(cfc_spec) -> ModelParams:
    """Flatten (Params, OBFParams, TBFParams) into the C struct."""
    mp, ob, tb = cfc_spec.model_params, cfc_spec.obf_params, \
        cfc_spec.tbf_params
    s = ModelParams()
    for k in ('lattice_depth', 'lattice_ratio', 'interaction_strength',
              'supercell_size', 'tbf_contact_cutoff', 'defect_magnitude',
              'well_width', 'barrier_width'):
        setattr(s, k, float(getattr(mp, k)))
    s.boson_number = int(mp.boson_number)
    s.defects_sep = int(mp.defects_sep)
    s.is_free = int(bool(mp.is_free))
    s.is_ideal = int(bool(mp.is_ideal))
    s.param_e0, s.param_k1, s.param_kp1 = (float(ob.param_e0),
                                           float(ob.param_k1),
                                           float(ob.param_kp1))
    s.param_k2, s.param_beta = float(tb.param_k2), float(tb.param_beta)
    s.param_r_off, s.param_am = float(tb.param_r_off), float(tb.param_am)
    return s


def _current_device():
    """Device index to bind to: LOCAL_RANK-aware through torch when a process
    group set the device, else 0."""
    try:
        import torch
        if torch.cuda.is_available():
            return torch.cuda.current_device()
    except Exception:     # torch missing or no GPU runtime
        pass
    return 0


class DeviceBuffer:
    """A plain fp64 array in HBM (qmc_buffer_*): inputs / outputs of
    `ModelEngine.evaluate_dev` that stay resident across calls."""

    def __init__(self, shape, device: t.Optional[int] = None):
        self._lib = _lib.load()
        self.shape = tuple(int(x) for x in np.atleast_1d(shape))
        self.nbytes = int(np.prod(self.shape)) * 8
        self.device = _current_device() if device is None else int(device)
        h = C.c_void_p()
        check(self._lib.qmc_buffer_alloc(self.device, self.nbytes, C.byref(h)))
        self._h = h

    @property
    def ptr(self):
        return self._h

    def upload(self, array):
        a = np.ascontiguousarray(array, dtype=np.float64)
        if a.shape != self.shape:
            raise ValueError(f'expected shape {self.shape}, got {a.shape}')
        check(self._lib.qmc_buffer_upload(self._h, a.ctypes.data, a.nbytes))
        return self

    def download(self):
        out = np.empty(self.shape)
        check(self._lib.qmc_buffer_download(out.ctypes.data, self._h,
                                            out.nbytes))
        return out

    def close(self):
        if getattr(self, '_h', None):
            self._lib.qmc_buffer_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ModelEngine:
    """Model constants bound to one GPU + stream (qmc_engine).

    `stream=None`: the engine creates its own non-blocking stream.
    `stream=<int>`: the caller's hipStream_t handle, used as it is -- 0 is the
    legacy default stream, which is what `torch.cuda.current_stream()
    .cuda_stream` returns unless a side stream is current.  Pass the stream
    torch is using whenever torch.distributed (RCCL) collectives must be
    ordered with the engine's kernels (`dist.DistributedDmc` checks it)."""

    def __init__(self, cfc_spec, device: t.Optional[int] = None,
                 stream: t.Optional[int] = None, fast_math: bool = False):
        self._lib = _lib.load()
        self.cfc_spec = cfc_spec
        self.num_particles = int(cfc_spec.model_params.boson_number)
        self.device = _current_device() if device is None else int(device)
        self._params = model_params_struct(cfc_spec)
        h = C.c_void_p()
        if stream is None:
            check(self._lib.qmc_engine_create(C.byref(self._params),
                                              self.device, None, C.byref(h)))
        else:
            check(self._lib.qmc_engine_create_on_stream(
                C.byref(self._params), self.device, C.c_void_p(int(stream)),
                C.byref(h)))
        self._h = h
        self.fast_math = False
        if fast_math:
            self.set_fast_math(True)

    def set_fast_math(self, on: bool) -> bool:
        """The reference's `jit_fastmath` knob: float pair loop (tables, sums
        over walkers, one-body factor, logs and the Metropolis test stay in
        double).  -> whether the variant is in effect for this model (it
        exists for boson_number > 32 and cutoffs not close to L/2)."""
        eff = C.c_int(0)
        check(self._lib.qmc_engine_set_fast_math(self._h, int(bool(on)),
                                                 C.byref(eff)))
        self.fast_math = bool(eff.value)
        return self.fast_math

    @property
    def stream_handle(self) -> int:
        """hipStream_t the engine launches on (0 = legacy default stream)."""
        st, owned = C.c_void_p(), C.c_int(0)
        check(self._lib.qmc_engine_stream(self._h, C.byref(st),
                                          C.byref(owned)))
        return int(st.value or 0)

    @property
    def owns_stream(self) -> bool:
        st, owned = C.c_void_p(), C.c_int(0)
        check(self._lib.qmc_engine_stream(self._h, C.byref(st),
                                          C.byref(owned)))
        return bool(owned.value)

    def profile_begin(self, max_launches: int = 4096):
        """Start timing every launch of the dominant kernel (one HIP event
        pair each, on the engine's stream)."""
        check(self._lib.qmc_engine_profile_begin(self._h, int(max_launches)))

    def profile_end(self):
        """-> (launches, total_ms, min_ms, max_ms); synchronises."""
        n = C.c_int64(0)
        tot, mn, mx = C.c_double(0), C.c_double(0), C.c_double(0)
        check(self._lib.qmc_engine_profile_end(self._h, C.byref(n),
                                               C.byref(tot), C.byref(mn),
                                               C.byref(mx)))
        return int(n.value), float(tot.value), float(mn.value), float(mx.value)

    def section_profile(self, reset: bool = True):
        """Diagnostic libraries only (built with -DQMC_TIMING,
        tools/section_times.py): {section name: (cycles, visits)} of wavefront
        lifetime per kernel section since the last reset; names ending in
        '@energy' are the sections inside the energy pass of the VMC step.
        Raises with the shipped library."""
        nsec = 32
        cyc = (C.c_uint64 * nsec)()
        vis = (C.c_uint64 * nsec)()
        check(self._lib.qmc_engine_section_profile(self._h, cyc, vis, nsec,
                                                   int(reset)))
        out = {}
        for i in range(nsec):
            if vis[i]:
                name = self._lib.qmc_section_name(i).decode()
                if i >= nsec // 2:
                    name += '@energy'
                out[name] = (int(cyc[i]), int(vis[i]))
        return out

    def section_cut(self, section: int):
        """Diagnostic libraries only (built with -DQMC_CUTS,
        tools/section_counts.py): end every wavefront of the walker kernels at
        section mark `section` (-1: never).  Raises with the shipped library."""
        check(self._lib.qmc_engine_section_cut(self._h, int(section)))

    def general_path_walkers(self, reset: bool = True) -> int:
        """Walker evaluations of the VMC / DMC stepping kernels that left the
        sorted-row pair sums (33 <= N <= 128) for the general one inside the
        same kernel since the last reset: 0 on equilibrated boxes, > 0 for
        clustered walkers (a device counter on the cold path only)."""
        out = (C.c_uint64 * 4)()
        check(self._lib.qmc_engine_diag_counters(self._h, out, 4, int(reset)))
        return int(out[0])

    def section_names(self):
        return [self._lib.qmc_section_name(i).decode() for i in range(16)]

    def close(self):
        if getattr(self, '_h', None):
            self._lib.qmc_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self._lib.qmc_engine_sync(self._h))

    def timer_start(self):
        check(self._lib.qmc_engine_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_float(0)
        check(self._lib.qmc_engine_timer_stop(self._h, C.byref(ms)))
        return float(ms.value)

    def evaluate(self, pos) -> EvalResult:
        """log|psi|, local energy, per-particle energy and drift of every
        configuration in pos[W, N]."""
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        if pos.ndim != 2 or pos.shape[1] != self.num_particles:
            raise ValueError('pos must have shape (W, boson_number)')
        W, n = pos.shape
        wf, en = np.zeros(W), np.zeros(W)
        ith, dr = np.zeros((W, n)), np.zeros((W, n))
        check(self._lib.qmc_evaluate(self._h, W, ptr(pos), ptr(wf), ptr(en),
                                     ptr(ith), ptr(dr)))
        return EvalResult(wf, en, ith, dr)

    def evaluate_dev(self, nconf, pos_ptr, wf_ptr=0, energy_ptr=0, ith_ptr=0,
                     drift_ptr=0):
        """Asynchronous evaluation on device-resident buffers (raw pointers,
        e.g. torch `tensor.data_ptr()`)."""
        check(self._lib.qmc_evaluate_dev(self._h, int(nconf), pos_ptr, wf_ptr,
                                         energy_ptr, ith_ptr, drift_ptr))


class VmcEnsemble:
    """W independent Metropolis chains resident on the GPU (qmc_vmc)."""

    def __init__(self, engine: ModelEngine, num_chains: int,
                 move_spread: float, rng_seed: int, chain0: int = 0,
                 gaussian: bool = False):
        self.engine = engine
        self._lib = engine._lib
        self.num_chains = int(num_chains)
        self.num_particles = engine.num_particles
        p = VmcParams(self.num_chains, float(move_spread),
                      int(rng_seed) & 0xFFFFFFFFFFFFFFFF, int(chain0),
                      int(bool(gaussian)))
        h = C.c_void_p()
        check(self._lib.qmc_vmc_create(engine._h, C.byref(p), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, '_h', None):
            self._lib.qmc_vmc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_state(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        if pos.shape != (self.num_chains, self.num_particles):
            raise ValueError('pos must have shape (num_chains, boson_number)')
        check(self._lib.qmc_vmc_set_state(self._h, ptr(pos)))

    def ssf_parts(self, num_modes: int) -> np.ndarray:
        """Mean over the chains of the S(k) parts (|rho_k|^2, Re rho_k,
        Im rho_k) of the current configurations, k_m = 2 pi m / L,
        m < num_modes -> [num_modes, 3]."""
        out = np.zeros((int(num_modes), 3))
        check(self._lib.qmc_vmc_ssf(self._h, int(num_modes), ptr(out)))
        return out / self.num_chains

    def get_state(self):
        """-> (pos[W, N], wf_abs_log[W], energy_carry[W])"""
        W, n = self.num_chains, self.num_particles
        pos, wf, ec = np.zeros((W, n)), np.zeros(W), np.zeros(W)
        check(self._lib.qmc_vmc_get_state(self._h, ptr(pos), ptr(wf), ptr(ec)))
        return pos, wf, ec

    def set_tape(self, tape):
        """TEST ONLY: tape[W, steps, N + 1]."""
        if tape is None:
            check(self._lib.qmc_vmc_set_tape(self._h, None, 0))
            return
        tape = np.ascontiguousarray(tape, dtype=np.float64)
        assert tape.ndim == 3 and tape.shape[0] == self.num_chains \
            and tape.shape[2] == self.num_particles + 1
        check(self._lib.qmc_vmc_set_tape(self._h, ptr(tape), tape.shape[1]))

    def run_block(self, nyield: int, sums: bool = True, series: bool = False,
                  confs: bool = False):
        """Advance every chain by `nyield` generator yields.
        -> dict with sum_energy, sum_energy2, num_accepted ([W]) and, if
        `series`, wf_abs_log / energy / move_stat ([nyield, W]) -- move_stat
        alone with series='stat' --; if `confs`, pos ([nyield, W, N])."""
        W = self.num_chains
        out = {}
        se = se2 = na = swf = sen = sst = spos = None
        if confs:
            spos = np.zeros((nyield, W, self.num_particles))
        if sums:
            se, se2 = np.zeros(W), np.zeros(W)
            na = np.zeros(W, dtype=np.int64)
        if series == 'stat':
            # (the move status alone: one byte per chain and yield)
            sst = np.zeros((nyield, W), dtype=np.uint8)
        elif series:
            swf, sen = np.zeros((nyield, W)), np.zeros((nyield, W))
            sst = np.zeros((nyield, W), dtype=np.uint8)
        check(self._lib.qmc_vmc_run_block(self._h, int(nyield), ptr(se),
                                          ptr(se2), ptr(na, _i64p), ptr(swf),
                                          ptr(sen), ptr(sst, _u8p),
                                          ptr(spos)))
        if confs:
            out.update(pos=spos)
        if sums:
            out.update(sum_energy=se, sum_energy2=se2, num_accepted=na)
        if series == 'stat':
            out.update(move_stat=sst.astype(bool))
        elif series:
            out.update(wf_abs_log=swf, energy=sen, move_stat=sst.astype(bool))
        return out

    def state_dev(self):
        """Device addresses (ints) of pos[W, N] and wf[W]."""
        a, b = C.c_void_p(), C.c_void_p()
        check(self._lib.qmc_vmc_state_dev(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def block_sums_dev(self):
        """Device addresses (ints) of sum_e[W], sum_e2[W], n_acc[W]."""
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(self._lib.qmc_vmc_block_sums_dev(self._h, C.byref(a),
                                               C.byref(b), C.byref(c)))
        return a.value, b.value, c.value


class DmcState(t.NamedTuple):
    confs: np.ndarray         # [maxw, 2, N]
    energy: np.ndarray        # [maxw]
    weight: np.ndarray        # [maxw]
    mask: np.ndarray          # [maxw] bool
    cloning_ref: np.ndarray   # [maxw] int64
    state_energy: float
    state_weight: float
    ref_energy: float
    accum_energy: float
    num_walkers: int


class DmcSeries(t.NamedTuple):
    energy: np.ndarray
    weight: np.ndarray
    num_walkers: np.ndarray   # uint64
    ref_energy: np.ndarray
    accum_energy: np.ndarray


class DmcEnsemble:
    """A walker population resident on the GPU (qmc_dmc)."""

    def __init__(self, engine: ModelEngine, time_step: float,
                 max_num_walkers: int, target_num_walkers: int,
                 num_walkers_control_factor: float, rng_seed: int,
                 slot0: int = 0, fix_stale_energy: bool = False,
                 external_reduce: bool = False):
        self.engine = engine
        self._lib = engine._lib
        self.max_num_walkers = int(max_num_walkers)
        self.num_particles = engine.num_particles
        p = DmcParams(self.max_num_walkers, int(target_num_walkers),
                      float(time_step), float(num_walkers_control_factor),
                      int(rng_seed) & 0xFFFFFFFFFFFFFFFF, int(slot0),
                      int(bool(fix_stale_energy)), int(bool(external_reduce)),
                      0)
        h = C.c_void_p()
        check(self._lib.qmc_dmc_create(engine._h, C.byref(p), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, '_h', None):
            self._lib.qmc_dmc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_state(self, pos, ref_energy: t.Optional[float] = None):
        """build_state semantics: energies and drifts are computed here."""
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        if pos.ndim != 2 or pos.shape[1] != self.num_particles:
            raise ValueError('pos must have shape (nw, boson_number)')
        check(self._lib.qmc_dmc_set_state(
            self._h, pos.shape[0], ptr(pos), int(ref_energy is not None),
            float(ref_energy if ref_energy is not None else 0.0)))

    def set_state_from_vmc(self, vmc: 'VmcEnsemble', num_walkers=None,
                           ref_energy: t.Optional[float] = None,
                           replicate: bool = False):
        """build_state from the current configurations of a VMC ensemble on
        the same engine, device to device (the first `num_walkers` chains;
        with `replicate` more walkers than chains are allowed and the chains
        are reused cyclically)."""
        nw = vmc.num_chains if num_walkers is None else int(num_walkers)
        if nw > vmc.num_chains and not replicate:
            raise ValueError('more walkers requested than chains')
        self.engine.sync()
        check(self._lib.qmc_dmc_set_state_from_vmc(
            self._h, vmc._h, nw, int(ref_energy is not None),
            float(ref_energy if ref_energy is not None else 0.0)))

    def set_full_state(self, confs, energy, weight, ref_energy: float,
                       slot_energy=None):
        confs = np.ascontiguousarray(confs, dtype=np.float64)
        energy = np.ascontiguousarray(energy, dtype=np.float64)
        weight = np.ascontiguousarray(weight, dtype=np.float64)
        nw = confs.shape[0]
        assert confs.shape == (nw, 2, self.num_particles)
        assert energy.shape == (nw,) and weight.shape == (nw,)
        if slot_energy is not None:
            slot_energy = np.ascontiguousarray(slot_energy, dtype=np.float64)
            assert slot_energy.shape == (self.max_num_walkers,)
        check(self._lib.qmc_dmc_set_full_state(self._h, nw, ptr(confs),
                                               ptr(energy), ptr(weight),
                                               ptr(slot_energy),
                                               float(ref_energy)))

    def set_tape(self, u, g, u_off, g_off):
        """TEST ONLY: recorded uniforms / standard normals + per-step offsets."""
        u = np.ascontiguousarray(u, dtype=np.float64)
        g = np.ascontiguousarray(g, dtype=np.float64)
        u_off = np.ascontiguousarray(u_off, dtype=np.int64)
        g_off = np.ascontiguousarray(g_off, dtype=np.int64)
        check(self._lib.qmc_dmc_set_tape(self._h, ptr(u), u.size, ptr(g),
                                         g.size, ptr(u_off, _i64p),
                                         ptr(g_off, _i64p), u_off.size))

    def run_block(self, nsteps: int, read: bool = True):
        """`nsteps` time steps; -> DmcSeries (or None when read=False: the
        series stay on the device until `read_series`)."""
        if not read:
            check(self._lib.qmc_dmc_run_block(self._h, int(nsteps), None, None,
                                              None, None, None))
            return None
        e, w = np.zeros(nsteps), np.zeros(nsteps)
        nw = np.zeros(nsteps, dtype=np.uint64)
        r, a = np.zeros(nsteps), np.zeros(nsteps)
        check(self._lib.qmc_dmc_run_block(self._h, int(nsteps), ptr(e), ptr(w),
                                          ptr(nw, _u64p), ptr(r), ptr(a)))
        return DmcSeries(e, w, nw, r, a)

    def set_estimators(self, num_modes=0, ssf_pure=False, ssf_pfw=1,
                       num_bins=0, dens_pure=False, dens_pfw=1):
        """Enable the S(k) / density estimators (0 disables one)."""
        self.num_modes, self.num_bins = int(num_modes), int(num_bins)
        p = DmcEstParams(self.num_modes, int(bool(ssf_pure)), int(ssf_pfw),
                         self.num_bins, int(bool(dens_pure)), int(dens_pfw))
        check(self._lib.qmc_dmc_set_estimators(self._h, C.byref(p)))

    def run_block_est(self, nsteps: int, eval_estimators: bool = True):
        """-> (DmcSeries, iter_ssf[nsteps, M, 3] or None,
        iter_density[nsteps, B, 1] or None)."""
        M, B = getattr(self, 'num_modes', 0), getattr(self, 'num_bins', 0)
        e, w = np.zeros(nsteps), np.zeros(nsteps)
        nw = np.zeros(nsteps, dtype=np.uint64)
        r, a = np.zeros(nsteps), np.zeros(nsteps)
        ssf = np.zeros((nsteps, M, 3)) if M else None
        dens = np.zeros((nsteps, B, 1)) if B else None
        check(self._lib.qmc_dmc_run_block_est(
            self._h, int(nsteps), int(bool(eval_estimators)), ptr(e), ptr(w),
            ptr(nw, _u64p), ptr(r), ptr(a), ptr(ssf), ptr(dens)))
        return DmcSeries(e, w, nw, r, a), ssf, dens

    def read_series(self, nsteps: int) -> DmcSeries:
        e, w = np.zeros(nsteps), np.zeros(nsteps)
        nw = np.zeros(nsteps, dtype=np.uint64)
        r, a = np.zeros(nsteps), np.zeros(nsteps)
        check(self._lib.qmc_dmc_read_series(self._h, int(nsteps), ptr(e),
                                            ptr(w), ptr(nw, _u64p), ptr(r),
                                            ptr(a)))
        return DmcSeries(e, w, nw, r, a)

    def get_state(self) -> DmcState:
        W, n = self.max_num_walkers, self.num_particles
        confs = np.zeros((W, 2, n))
        energy, weight = np.zeros(W), np.zeros(W)
        mask = np.zeros(W, dtype=np.uint8)
        ref = np.zeros(W, dtype=np.int64)
        sc = np.zeros(5)
        check(self._lib.qmc_dmc_get_state(self._h, ptr(confs), ptr(energy),
                                          ptr(weight), ptr(mask, _u8p),
                                          ptr(ref, _i64p), ptr(sc)))
        return DmcState(confs, energy, weight, mask.astype(bool), ref,
                        float(sc[0]), float(sc[1]), float(sc[2]),
                        float(sc[3]), int(sc[4]))

    def get_scalars(self):
        """-> (state_energy, state_weight, ref_energy, accum_energy,
        num_walkers) without downloading the population."""
        sc = np.zeros(5)
        check(self._lib.qmc_dmc_get_state(self._h, None, None, None, None,
                                          None, ptr(sc)))
        return (float(sc[0]), float(sc[1]), float(sc[2]), float(sc[3]),
                int(sc[4]))

    # -- multi-GPU building blocks ------------------------------------------
    def step_local(self, partial_ptr: int):
        check(self._lib.qmc_dmc_step_local(self._h, partial_ptr))

    def step_finish(self, total_ptr: int):
        check(self._lib.qmc_dmc_step_finish(self._h, total_ptr))

    def num_walkers(self) -> int:
        n = C.c_int64(0)
        check(self._lib.qmc_dmc_num_walkers(self._h, C.byref(n)))
        return int(n.value)

    def walker_record_size(self) -> int:
        """Doubles per walker record of export / import (3N + 2 + the
        forward-walking estimator rows when estimators are set)."""
        n = C.c_int64(0)
        check(self._lib.qmc_dmc_walker_record_size(self._h, C.byref(n)))
        return int(n.value)

    def export_walkers(self, first: int, count: int, buf_ptr: int):
        check(self._lib.qmc_dmc_export_walkers(self._h, int(first), int(count),
                                               buf_ptr))

    def import_walkers(self, count: int, buf_ptr: int):
        """Append at the device's population count (synchronises)."""
        check(self._lib.qmc_dmc_import_walkers(self._h, int(count), buf_ptr))

    def import_walkers_at(self, first: int, count: int, buf_ptr: int):
        """Stream-ordered: records -> slots [first, first + count); the
        population size becomes first + count."""
        check(self._lib.qmc_dmc_import_walkers_at(self._h, int(first),
                                                  int(count), buf_ptr))

    def set_num_walkers(self, nw: int):
        """Stream-ordered truncation (the caller knows the population size)."""
        check(self._lib.qmc_dmc_set_num_walkers(self._h, int(nw)))

    def truncate(self, new_nw: int):
        check(self._lib.qmc_dmc_truncate(self._h, int(new_nw)))

    # -- estimators of a split-step (multi-GPU) run -------------------------
    def est_begin_block(self, nsteps: int):
        check(self._lib.qmc_dmc_est_begin_block(self._h, int(nsteps)))

    def step_estimators(self, step_idx: int):
        check(self._lib.qmc_dmc_step_estimators(self._h, int(step_idx)))

    def est_iter_dev(self):
        """Device addresses (ints or None) of this rank's iter_ssf /
        iter_density rows of the open estimator block."""
        a, b = C.c_void_p(), C.c_void_p()
        check(self._lib.qmc_dmc_est_iter_dev(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value
