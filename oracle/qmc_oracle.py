"""ctypes front-end of the CPU oracle (libqmc_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by `__graft_entry__.smoke()` and
by the `cpu_baseline` leg of bench.py -- never by the product package.  The
algorithm lives in qmc_oracle.c (a restatement of the reference, pinned by
tests/golden/); this file only marshals numpy arrays.
"""
import ctypes as C
import os
import subprocess
from math import sqrt

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libqmc_oracle.so')

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_i64p = C.POINTER(C.c_int64)


class OrcModel(C.Structure):
    _fields_ = [
        ('lattice_depth', C.c_double), ('lattice_ratio', C.c_double),
        ('interaction_strength', C.c_double), ('boson_number', C.c_int64),
        ('supercell_size', C.c_double), ('tbf_contact_cutoff', C.c_double),
        ('defect_magnitude', C.c_double), ('defects_sep', C.c_int64),
        ('well_width', C.c_double), ('barrier_width', C.c_double),
        ('is_free', C.c_int64), ('is_ideal', C.c_int64),
        ('param_e0', C.c_double), ('param_k1', C.c_double),
        ('param_kp1', C.c_double), ('param_k2', C.c_double),
        ('param_beta', C.c_double), ('param_r_off', C.c_double),
        ('param_am', C.c_double),
    ]


class OrcVmcCfg(C.Structure):
    _fields_ = [('move_spread', C.c_double), ('seed', C.c_uint64),
                ('chain', C.c_uint32), ('step0', C.c_uint32),
                ('yield_initial', C.c_int32), ('gaussian', C.c_int32)]


class OrcDmcCfg(C.Structure):
    _fields_ = [('max_num_walkers', C.c_int64),
                ('target_num_walkers', C.c_int64),
                ('time_step', C.c_double), ('control_factor', C.c_double),
                ('seed', C.c_uint64), ('slot0', C.c_uint32),
                ('fix_stale_energy', C.c_int32)]


class OrcDmcState(C.Structure):
    _fields_ = [('prev_confs', _dp), ('prev_energy', _dp), ('prev_weight', _dp),
                ('actual_confs', _dp), ('actual_energy', _dp),
                ('actual_weight', _dp),
                ('next_confs', _dp), ('next_energy', _dp), ('next_weight', _dp),
                ('actual_mask', _u8p), ('cloning_ref', _i64p),
                ('prev_num_walkers', C.c_int64), ('ref_energy', C.c_double),
                ('total_energy', C.c_double), ('total_weight', C.c_double),
                ('step', C.c_uint32)]


class OrcDmcYield(C.Structure):
    _fields_ = [('energy', C.c_double), ('weight', C.c_double),
                ('num_walkers', C.c_int64), ('ref_energy', C.c_double),
                ('accum_energy', C.c_double), ('n_uniform', C.c_int64),
                ('n_normal', C.c_int64)]


def build(force=False):
    """Compile libqmc_oracle.so with the committed Makefile."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) <
            os.path.getmtime(os.path.join(_HERE, 'qmc_oracle.c'))):
        subprocess.check_call(['make', '-C', _HERE, '-B' if force else '-s'])
    return _LIB_PATH


_lib = None
_active_path = _LIB_PATH
_native_dir = None
# (the TIMED build: libm builtins allowed -- numba's LLVM inlines fabs /
# copysign / sqrt too; the bit-exact checker of the Makefile keeps -fno-builtin)
NATIVE_FLAGS = '-O3 -march=native -fPIC -fopenmp -ffp-contract=off -std=gnu11'


def build_native():
    """For the `cpu_baseline` leg of bench.py: rebuild the oracle ON THE BOX
    THAT RUNS IT with -O3 -march=native (the committed Makefile
    builds the bit-exact checker with -O2 and no target flags, and a
    -march=native object must not travel between machines) and make it the
    library `lib()` loads from now on.  Still no -ffast-math and no FMA
    contraction: the reference default is jit_fastmath=False.
    -> the flag string of the build in use."""
    global _lib, _active_path, _native_dir
    import tempfile
    # a private directory (mode 0700, fresh name): nobody else can put a
    # library where this process is about to load one from
    if _native_dir is None:
        import atexit
        import shutil
        _native_dir = tempfile.mkdtemp(prefix='qmc_oracle_native_')
        atexit.register(shutil.rmtree, _native_dir, ignore_errors=True)
    out = os.path.join(_native_dir, 'libqmc_oracle_native.so')
    cc = os.environ.get('CC', 'gcc')
    cmd = [cc] + NATIVE_FLAGS.split() + ['-shared', '-o', out,
                                         os.path.join(_HERE, 'qmc_oracle.c'),
                                         '-lm']
    try:
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        build()
        return 'gcc -O2 (oracle/Makefile; native rebuild failed)'
    _active_path = out
    _lib = None
    return f'{cc} {NATIVE_FLAGS}'


def lib():
    global _lib
    if _lib is None:
        if _active_path == _LIB_PATH:
            build()
        L = C.CDLL(_active_path)
        L.orc_wf_abs_log.restype = C.c_double
        L.orc_wf_abs_log.argtypes = [C.POINTER(OrcModel), _dp]
        L.orc_energy_drift.restype = C.c_double
        L.orc_energy_drift.argtypes = [C.POINTER(OrcModel), _dp, _dp, _dp]
        L.orc_np_sum.restype = C.c_double
        L.orc_np_sum.argtypes = [_dp, C.c_int64]
        L.orc_vmc_chain.restype = C.c_int64
        L.orc_vmc_chain.argtypes = [C.POINTER(OrcModel), C.POINTER(OrcVmcCfg),
                                    _dp, _dp, _dp, C.c_int64, _dp, _dp, _dp,
                                    _u8p]
        L.orc_vmc_ensemble.restype = None
        L.orc_vmc_ensemble.argtypes = [C.POINTER(OrcModel),
                                       C.POINTER(OrcVmcCfg), C.c_int64, _dp,
                                       _dp, _dp, C.c_int64, _dp, _dp, _i64p,
                                       C.c_int]
        L.orc_dmc_step.restype = None
        L.orc_dmc_step.argtypes = [C.POINTER(OrcModel), C.POINTER(OrcDmcCfg),
                                   C.POINTER(OrcDmcState), _dp, _dp,
                                   C.POINTER(OrcDmcYield), C.c_int]
        L.orc_dmc_prepare.restype = None
        L.orc_dmc_prepare.argtypes = [C.POINTER(OrcModel), C.c_int64, _dp, _dp,
                                      C.c_int]
        L.orc_philox_uniform2.restype = None
        L.orc_philox_uniform2.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32,
                                          C.c_uint32, C.c_uint32, _dp]
        L.orc_dmc_normal.restype = C.c_double
        L.orc_dmc_normal.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32,
                                     C.c_uint32]
        L.orc_philox_normal.restype = C.c_double
        L.orc_philox_normal.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.c_uint32]
        _u32p = C.POINTER(C.c_uint32)
        L.orc_philox2x32.restype = None
        L.orc_philox2x32.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, _u32p]
        L.orc_vmc_move_block.restype = None
        L.orc_vmc_move_block.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32,
                                         C.c_uint32, _u32p]
        L.orc_vmc_move_unit.restype = C.c_double
        L.orc_vmc_move_unit.argtypes = [C.c_uint32]
        L.orc_vmc_accept_uniform.restype = C.c_double
        L.orc_vmc_accept_uniform.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(a, typ=_dp):
    return None if a is None else a.ctypes.data_as(typ)


def model_from_params(params, obf, tbf):
    """OrcModel from Params / OBFParams / TBFParams tuples or dicts."""
    def g(o, k):
        return o[k] if isinstance(o, dict) else getattr(o, k)
    m = OrcModel()
    for k in ('lattice_depth', 'lattice_ratio', 'interaction_strength',
              'supercell_size', 'tbf_contact_cutoff', 'defect_magnitude',
              'well_width', 'barrier_width'):
        setattr(m, k, float(g(params, k)))
    m.boson_number = int(g(params, 'boson_number'))
    m.defects_sep = int(g(params, 'defects_sep'))
    m.is_free = int(bool(g(params, 'is_free')))
    m.is_ideal = int(bool(g(params, 'is_ideal')))
    for k in ('param_e0', 'param_k1', 'param_kp1'):
        setattr(m, k, float(g(obf, k)))
    for k in ('param_k2', 'param_beta', 'param_r_off', 'param_am'):
        setattr(m, k, float(g(tbf, k)))
    return m


def model_from_cfc(cfc_spec):
    return model_from_params(cfc_spec.model_params, cfc_spec.obf_params,
                             cfc_spec.tbf_params)


def max_threads():
    return int(lib().orc_max_threads())


def wf_abs_log(model, pos):
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    return float(lib().orc_wf_abs_log(C.byref(model), _p(pos)))


def energy_drift(model, pos):
    """-> (energy, ith_energy[N], drift[N])"""
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    n = model.boson_number
    ie, fd = np.zeros(n), np.zeros(n)
    e = lib().orc_energy_drift(C.byref(model), _p(pos), _p(ie), _p(fd))
    return float(e), ie, fd


def evaluate_set(model, pos_set):
    """Batch version over pos_set[W, N] -> (wf[W], energy[W], ith[W,N],
    drift[W,N])."""
    pos_set = np.ascontiguousarray(pos_set, dtype=np.float64)
    W, n = pos_set.shape
    wf, en = np.zeros(W), np.zeros(W)
    ie, fd = np.zeros((W, n)), np.zeros((W, n))
    for w in range(W):
        wf[w] = wf_abs_log(model, pos_set[w])
        en[w], ie[w], fd[w] = energy_drift(model, pos_set[w])
    return wf, en, ie, fd


def np_sum(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return float(lib().orc_np_sum(_p(a), a.size))


def philox_uniform2(seed, slot, step, index, stream):
    u = np.zeros(2)
    lib().orc_philox_uniform2(seed, slot, step, index, stream, _p(u))
    return u


def philox2x32(c0, c1, key):
    """Philox2x32-10 block of counter (c0, c1) under `key` -> (w0, w1)."""
    w = (C.c_uint32 * 2)()
    lib().orc_philox2x32(c0, c1, key, w)
    return int(w[0]), int(w[1])


def vmc_move_block(seed, slot, step, index):
    """The VMC move stream's block of (seed; chain slot, step, particle)
    -> (w0, w1)."""
    w = (C.c_uint32 * 2)()
    lib().orc_vmc_move_block(seed, slot, step, index, w)
    return int(w[0]), int(w[1])


def vmc_move_unit(w0):
    """displacement / move_spread of a particle from word 0 of its block"""
    return float(lib().orc_vmc_move_unit(w0))


def vmc_accept_uniform(w1_p0, w1_p1):
    return float(lib().orc_vmc_accept_uniform(w1_p0, w1_p1))


def dmc_normal(seed, slot, step, index):
    """The standard normal that moves particle `index` of walker slot `slot`
    at DMC time step `step` (Philox2x32-10 block of the step pair)."""
    return float(lib().orc_dmc_normal(seed, slot, step, index))


def philox_normal(seed, slot, step, index, stream):
    return float(lib().orc_philox_normal(seed, slot, step, index, stream))


class VmcChain:
    """One Markov chain with the yield protocol of the reference's
    `states_generator` (qmc_base/vmc.py:571-646)."""

    def __init__(self, model, pos, move_spread, seed=0, chain=0,
                 gaussian=False):
        self.model = model
        self.pos = np.array(pos, dtype=np.float64)
        self.wf = np.array([wf_abs_log(model, self.pos)])
        self.e_prev = np.zeros(1)
        self.cfg = OrcVmcCfg(float(move_spread), int(seed), int(chain), 0, 1,
                             int(bool(gaussian)))

    def run(self, nyield, tape=None):
        """-> (wf[nyield], energy[nyield], move_stat[nyield], n_accepted)"""
        n = self.model.boson_number
        owf, oen = np.zeros(nyield), np.zeros(nyield)
        ost = np.zeros(nyield, dtype=np.uint8)
        if tape is not None:
            tape = np.ascontiguousarray(tape, dtype=np.float64)
            need = (nyield - self.cfg.yield_initial) * (n + 1)
            assert tape.size >= need, (tape.size, need)
        acc = lib().orc_vmc_chain(C.byref(self.model), C.byref(self.cfg),
                                  _p(self.pos), _p(self.wf), _p(self.e_prev),
                                  nyield, _p(tape), _p(owf), _p(oen),
                                  _p(ost, _u8p))
        real = nyield - self.cfg.yield_initial
        self.cfg.step0 += real
        self.cfg.yield_initial = 0
        return owf, oen, ost.astype(bool), int(acc)


def vmc_ensemble(model, pos, wf, e_prev, move_spread, seed, nyield,
                 step0=0, yield_initial=False, chain0=0, nthreads=None):
    """Advance an ensemble of chains in place; -> (sum_e, sum_e2, n_acc)."""
    W = pos.shape[0]
    cfg = OrcVmcCfg(float(move_spread), int(seed), int(chain0), int(step0),
                    int(bool(yield_initial)), 0)
    se, se2 = np.zeros(W), np.zeros(W)
    na = np.zeros(W, dtype=np.int64)
    nthreads = nthreads or max_threads()
    lib().orc_vmc_ensemble(C.byref(model), C.byref(cfg), W, _p(pos), _p(wf),
                           _p(e_prev), nyield, _p(se), _p(se2), _p(na, _i64p),
                           nthreads)
    return se, se2, na


class DmcEnsemble:
    """Walker ensemble with the three-buffer scheme and yield protocol of the
    reference's DMC `states_generator` (qmc_base/dmc.py:679-785), started from
    `build_state` semantics (mrbp_qmc/dmc.py:268-328)."""

    def __init__(self, model, ini_pos, time_step, max_num_walkers,
                 target_num_walkers, control_factor, seed=0, ref_energy=None,
                 slot0=0, fix_stale_energy=False, nthreads=1):
        self.model = model
        n = model.boson_number
        ini_pos = np.asarray(ini_pos, dtype=np.float64)[-target_num_walkers:]
        nw = len(ini_pos)
        maxw = int(max_num_walkers)
        self.nthreads = nthreads
        confs = np.zeros((maxw, 2, n))
        confs[:nw, 0, :] = ini_pos
        energy = np.zeros(maxw)
        weight = np.zeros(maxw)
        lib().orc_dmc_prepare(C.byref(model), nw, _p(confs), _p(energy),
                              nthreads)
        weight[:nw] = 1.
        e_sum = float((energy[:nw] * weight[:nw]).sum())
        w_sum = float(weight[:nw].sum())
        self.ini_energy = energy.copy()
        self.ini_confs = confs.copy()
        if ref_energy is None:
            ref_energy = e_sum / w_sum
        self.bufs = dict(
            prev_confs=confs.copy(), prev_energy=energy.copy(),
            prev_weight=weight.copy(),
            actual_confs=confs.copy(), actual_energy=energy.copy(),
            actual_weight=weight.copy(),
            next_confs=confs.copy(), next_energy=energy.copy(),
            next_weight=weight.copy(),
            actual_mask=np.zeros(maxw, dtype=np.uint8),
            cloning_ref=np.zeros(maxw, dtype=np.int64))
        st = OrcDmcState()
        for k, v in self.bufs.items():
            typ = _u8p if k == 'actual_mask' else \
                _i64p if k == 'cloning_ref' else _dp
            setattr(st, k, _p(v, typ))
        st.prev_num_walkers = nw
        st.ref_energy = float(ref_energy)
        st.total_energy = 0.
        st.total_weight = 0.
        st.step = 0
        self.st = st
        self.cfg = OrcDmcCfg(maxw, int(target_num_walkers), float(time_step),
                             float(control_factor), int(seed), int(slot0),
                             int(bool(fix_stale_energy)))
        self.sigma = sqrt(2 * time_step)

    def step(self, u_tape=None, g_tape=None):
        out = OrcDmcYield()
        if u_tape is not None:
            u_tape = np.ascontiguousarray(u_tape, dtype=np.float64)
        if g_tape is not None:
            g_tape = np.ascontiguousarray(g_tape, dtype=np.float64)
        lib().orc_dmc_step(C.byref(self.model), C.byref(self.cfg),
                           C.byref(self.st), _p(u_tape), _p(g_tape),
                           C.byref(out), self.nthreads)
        return out

    # views of the yielded ("actual") state
    @property
    def confs(self):
        return self.bufs['actual_confs']

    @property
    def energy(self):
        return self.bufs['actual_energy']

    @property
    def cloning_ref(self):
        return self.bufs['cloning_ref']


class DmcEstimators:
    """numpy restatement of the reference's DMC estimators evaluated on the
    yielded states of a block:

    * static structure factor parts |rho_k|^2, Re rho_k, Im rho_k, mixed or
      pure / forward walking (qmc_base/jastrow/dmc.py:363-631; per-block
      resets qmc_base/dmc.py:897-909),
    * density histogram, mixed or pure (mrbp_qmc/dmc.py:472-547,
      qmc_base/jastrow/dmc.py:195-302) -- including the reference's own
      behaviour that the mixed histogram keeps accumulating in the two
      alternating per-slot buffers and that the pure one is transported by
      slot index, not through the cloning table.

    ssf / dens = (num, as_pure_est, pfw_num_time_steps) or None.
    """

    def __init__(self, supercell_size, num_particles, max_num_walkers,
                 num_time_steps_block, ssf=None, dens=None):
        self.L = float(supercell_size)
        self.n = int(num_particles)
        self.maxw = int(max_num_walkers)
        self.nts = int(num_time_steps_block)
        self.ssf = ssf
        self.dens = dens
        if ssf is not None:
            self.momenta = np.arange(ssf[0]) * 2 * np.pi / self.L
        self.reset_block()

    def reset_block(self):
        if self.ssf is not None:
            m = self.ssf[0]
            self.iter_ssf = np.zeros((self.nts, m, 3))
            self.aux_ssf = np.zeros((2, self.maxw, m, 3))
        if self.dens is not None:
            b = self.dens[0]
            self.iter_density = np.zeros((self.nts, b, 1))
            self.aux_density = np.zeros((2, self.maxw, b, 1))

    def _ssf_parts(self, pos):
        ph = self.momenta[:, None] * pos[None, :]
        re, im = np.cos(ph).sum(axis=1), np.sin(ph).sum(axis=1)
        return np.stack([re * re + im * im, re, im], axis=1)

    def step(self, step_idx, confs, num_walkers, cloning_ref):
        nw = int(num_walkers)
        prev_i, act_i = step_idx % 2 - 1, step_idx % 2
        if self.ssf is not None:
            m, pure, pfw = self.ssf
            prev, act = self.aux_ssf[prev_i], self.aux_ssf[act_i]
            for s in range(nw):
                if not pure:
                    act[s] = self._ssf_parts(confs[s, 0])
                elif step_idx >= pfw:
                    act[s] = prev[cloning_ref[s]]
                else:
                    act[s] = self._ssf_parts(confs[s, 0]) + prev[cloning_ref[s]]
            div = 1 if not pure else (step_idx + 1 if step_idx < pfw else pfw)
            self.iter_ssf[step_idx] = act[:nw].sum(axis=0)
            if pure:
                self.iter_ssf[step_idx] /= div
        if self.dens is not None:
            b, pure, pfw = self.dens
            prev, act = self.aux_density[prev_i], self.aux_density[act_i]
            bin_size = self.L / b
            if pure:
                act[:] = prev
            for s in range(nw):
                if pure and step_idx >= pfw:
                    continue
                for z in confs[s, 0]:
                    act[s, int(z // bin_size), 0] += 1
            div = 1 if not pure else (step_idx + 1 if step_idx < pfw else pfw)
            self.iter_density[step_idx] = act[:nw].sum(axis=0)
            if pure:
                self.iter_density[step_idx] /= div
