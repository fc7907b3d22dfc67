"""DMC sampling of the Bloch-Phonon model on the GPU.

`Sampling` keeps the reference surface (mrbp_qmc/dmc.py:143-334): the same
attrs fields and defaults, `build_state`, `states`, `blocks`,
`state_data_blocks`, `cfc_spec`, `ddf_params`.  The population lives in HBM
behind a `qmc_dmc` handle; a block of time steps is enqueued without a host
round trip and only the per-step series come back.

`jit_parallel` is accepted for drop-in compatibility (there is no numba here);
`jit_fastmath` selects the engine's reduced-precision pair loop (float), as in
the reference only together with `jit_parallel`.  `fix_stale_energy` is an extension: False
reproduces the reference's branching weight, True uses the
parent's energy.  The density / S(k) estimators (mixed and pure /
forward-walking) run on the device after every kept
time step and reproduce the reference's semantics, per-block resets included.
"""
import typing as t
from math import pi, sqrt

import attr
import numpy as np

from .. import utils
from ..engine import DmcEnsemble, ModelEngine
from ..qmc_base import dmc as dmc_base
from . import model

__all__ = ['CFCSpec', 'DDFParams', 'DensityEstSpec', 'Sampling', 'SSFEstSpec',
           'State', 'StateError']

State = dmc_base.State


class StateError(ValueError):
    """Flags errors related to the handling of a DMC state."""


class DDFParams(t.NamedTuple):
    """Diffusion-and-drift parameters (mrbp_qmc/dmc.py:56-62)."""
    boson_number: int
    time_step: float
    sigma_spread: float
    lower_bound: float
    upper_bound: float


class DensityParams(t.NamedTuple):
    num_bins: int
    as_pure_est: bool
    pfw_num_time_steps: int
    assume_none: bool


class SSFParams(t.NamedTuple):
    num_modes: int
    as_pure_est: bool
    pfw_num_time_steps: int
    assume_none: bool


class CFCSpec(t.NamedTuple):
    model_params: model.Params
    obf_params: model.OBFParams
    tbf_params: model.TBFParams
    ddf_params: DDFParams
    density_params: t.Optional[DensityParams]
    ssf_params: t.Optional[SSFParams] = None


@attr.s(auto_attribs=True, frozen=True)
class DensityEstSpec:
    """mrbp_qmc/dmc.py:103-122."""
    num_bins: int
    as_pure_est: bool = True
    pfw_num_time_steps: int = 99999999


@attr.s(auto_attribs=True, frozen=True)
class SSFEstSpec:
    """mrbp_qmc/dmc.py:125-140."""
    num_modes: int
    as_pure_est: bool = True
    pfw_num_time_steps: t.Optional[int] = None

    def __attrs_post_init__(self):
        if self.pfw_num_time_steps is None:
            object.__setattr__(self, 'pfw_num_time_steps', 99999999)


@attr.s(auto_attribs=True, frozen=True)
class Sampling:
    """A class to realize a DMC sampling (mrbp_qmc/dmc.py:143-160)."""

    model_spec: model.Spec
    time_step: float
    max_num_walkers: int
    target_num_walkers: int
    num_walkers_control_factor: t.Optional[float] = None
    rng_seed: t.Optional[int] = None
    density_est_spec: t.Optional[DensityEstSpec] = None
    ssf_est_spec: t.Optional[SSFEstSpec] = None
    jit_parallel: bool = True
    jit_fastmath: bool = False
    fix_stale_energy: bool = False

    def __attrs_post_init__(self):
        if self.rng_seed is None:
            object.__setattr__(self, 'rng_seed',
                               int(utils.get_random_rng_seed()))
        if self.num_walkers_control_factor is None:
            object.__setattr__(self, 'num_walkers_control_factor', 1.25e-1)

    # -- parameters (mrbp_qmc/dmc.py:172-236) ---------------------------------
    @property
    def ddf_params(self) -> DDFParams:
        z_min, z_max = self.model_spec.boundaries
        return DDFParams(self.model_spec.boson_number, self.time_step,
                         sqrt(2 * self.time_step), z_min, z_max)

    @property
    def density_params(self) -> DensityParams:
        """mrbp_qmc/dmc.py:187-204."""
        d = self.density_est_spec
        if d is None:
            return DensityParams(1, False, 1, True)
        return DensityParams(d.num_bins, d.as_pure_est, d.pfw_num_time_steps,
                             False)

    @property
    def ssf_params(self) -> SSFParams:
        """mrbp_qmc/dmc.py:206-225."""
        f = self.ssf_est_spec
        if f is None:
            return SSFParams(1, False, 1, True)
        return SSFParams(f.num_modes, f.as_pure_est, f.pfw_num_time_steps,
                         False)

    @property
    def density_bins_edges(self) -> np.ndarray:
        if self.density_est_spec is None:
            raise TypeError('the density spec has no been specified')
        return np.linspace(0, self.model_spec.supercell_size,
                           self.density_est_spec.num_bins + 1)

    @property
    def cfc_spec(self) -> CFCSpec:
        ms = self.model_spec
        return CFCSpec(ms.params, ms.obf_params, ms.tbf_params,
                       self.ddf_params, self.density_params, self.ssf_params)

    @property
    def state_confs_shape(self):
        return (self.max_num_walkers,) + self.model_spec.sys_conf_shape

    @property
    def state_props_shape(self):
        return self.max_num_walkers,

    @property
    def core_funcs(self):
        return model.core_funcs

    # -- engine plumbing --------------------------------------------------------
    def set_replay_tape(self, uniform, normal, n_uniform, n_normal):
        """TEST ONLY: the next generator replays recorded streams instead of
        Philox: the uniforms of the branching loop and the standard normals
        of the diffusion, with the number of draws of every time step."""
        if uniform is None:
            object.__setattr__(self, '_replay_tape', None)
            return
        u_off = np.concatenate([[0], np.cumsum(n_uniform)[:-1]])
        g_off = np.concatenate([[0], np.cumsum(n_normal)[:-1]])
        object.__setattr__(self, '_replay_tape',
                           (np.asarray(uniform, dtype=np.float64),
                            np.asarray(normal, dtype=np.float64),
                            u_off.astype(np.int64), g_off.astype(np.int64)))

    @property
    def fast_math_in_effect(self) -> bool:
        """`jit_fastmath` asks for the reduced-precision pair loop; as in the
        reference it only takes effect together with `jit_parallel`
        (mrbp_qmc/dmc.py:647-654: the (False, True) entry of the core
        functions table is built without fastmath)."""
        return bool(self.jit_parallel and self.jit_fastmath)

    def _new_ensemble(self, target=None, slot0=0, device=None, stream=None,
                      external_reduce=False):
        if self.fast_math_in_effect:
            # in the reference `fastmath` relaxes IEEE rules but still computes
            # in double; here it changes the precision CLASS of the pair loop:
            # say so every time
            import warnings
            warnings.warn(
                'jit_fastmath=True with jit_parallel=True runs the pair loop '
                'in float (fp32) on the device: local energy ~1e-7 relative, '
                'drift of a pair closer than ~1e-6 L loses its leading digits '
                '(DESIGN.md section 4, "Reduced precision"); the DMC result '
                'with this option is not pinned against the reference',
                RuntimeWarning, stacklevel=3)
        eng = ModelEngine(self.model_spec.cfc_spec, device=device,
                          stream=stream, fast_math=self.fast_math_in_effect)
        ens = DmcEnsemble(eng, self.time_step, self.max_num_walkers,
                          target if target is not None
                          else self.target_num_walkers,
                          self.num_walkers_control_factor, self.rng_seed,
                          slot0=slot0, fix_stale_energy=self.fix_stale_energy,
                          external_reduce=external_reduce)
        return eng, ens

    def _to_state(self, s) -> State:
        props = dmc_base.StateProps(s.energy, s.weight, s.mask)
        bspec = dmc_base.BranchingSpec(
            np.zeros(self.max_num_walkers, dtype=np.int64), s.cloning_ref)
        return State(s.confs, props, s.state_energy, s.state_weight,
                     s.num_walkers, s.ref_energy, s.accum_energy,
                     self.max_num_walkers, bspec)

    def build_state(self, sys_conf_set: np.ndarray,
                    ref_energy: t.Optional[float] = None) -> State:
        """mrbp_qmc/dmc.py:268-328: keep the last `target_num_walkers`
        configurations, evaluate energy + drift of each, unit weights, initial
        E_ref = mean energy unless given."""
        sys_conf_set = np.asarray(sys_conf_set)
        confs_shape = self.state_confs_shape
        if len(confs_shape) == len(sys_conf_set.shape):
            if confs_shape[1:] != sys_conf_set.shape[1:]:
                raise StateError("sys_conf_set is not a valid set of "
                                 "configurations of the model spec")
        sys_conf_set = sys_conf_set[-self.target_num_walkers:]
        nw = len(sys_conf_set)
        if nw > self.max_num_walkers:
            raise StateError('more initial configurations than '
                             'max_num_walkers')
        eng, ens = self._new_ensemble()
        try:
            ens.set_state(sys_conf_set[:, model.SysConfSlot.pos, :],
                          ref_energy)
            s = ens.get_state()
        finally:
            ens.close()
            eng.close()
        e = s.energy[:nw]
        w = s.weight[:nw]
        state_energy = float((e * w).sum())
        state_weight = float(w.sum())
        energy = state_energy / state_weight
        props = dmc_base.StateProps(s.energy, s.weight, s.mask)
        bspec = dmc_base.BranchingSpec(
            np.zeros(self.max_num_walkers, dtype=np.int64),
            np.zeros(self.max_num_walkers, dtype=np.int64))
        return State(s.confs, props, state_energy, state_weight, nw,
                     energy if ref_energy is None else ref_energy, energy,
                     self.max_num_walkers, bspec)

    def _start(self, ini_state: State, target=None):
        if ini_state.max_num_walkers != self.max_num_walkers:
            raise StateError('the state was built for a different '
                             'max_num_walkers')
        eng, ens = self._new_ensemble(target=target)
        nw = int(ini_state.num_walkers)
        ens.set_full_state(ini_state.confs[:nw], ini_state.props.energy[:nw],
                           ini_state.props.weight[:nw], ini_state.ref_energy,
                           slot_energy=ini_state.props.energy)
        tape = getattr(self, '_replay_tape', None)
        if tape is not None:
            ens.set_tape(*tape)
        return eng, ens

    # -- generators (qmc_base/dmc.py:311-364) ---------------------------------
    def states(self, ini_state: State) -> t.Iterator[State]:
        """One State per time step (the post-branching population)."""
        eng, ens = self._start(ini_state)
        try:
            while True:
                ens.run_block(1, read=False)
                yield self._to_state(ens.get_state())
        finally:
            ens.close()
            eng.close()

    def blocks(self, ini_state: State, num_time_steps_block: int,
               burn_in_blocks: int) -> t.Iterator[dmc_base.SamplingBlock]:
        """qmc_base/dmc.py:831-969.  `burn_in_blocks` only gates the
        estimators in the reference; the propagation is identical."""
        nts = int(num_time_steps_block)
        eng, ens = self._start(ini_state)
        dp, sp = self.density_params, self.ssf_params
        with_est = not (dp.assume_none and sp.assume_none)
        if with_est:
            ens.set_estimators(
                0 if sp.assume_none else sp.num_modes, sp.as_pure_est,
                sp.pfw_num_time_steps,
                0 if dp.assume_none else dp.num_bins, dp.as_pure_est,
                dp.pfw_num_time_steps)
        block_idx = 0
        try:
            while True:
                iter_ssf = iter_density = None
                if with_est:
                    # estimators only once the burn-in blocks are over
                    # (qmc_base/dmc.py:916, 928)
                    ser, iter_ssf, iter_density = ens.run_block_est(
                        nts, block_idx >= burn_in_blocks)
                else:
                    ser = ens.run_block(nts)
                props = dmc_base.PropsData(ser.energy, ser.weight,
                                           ser.num_walkers, ser.ref_energy,
                                           ser.accum_energy)
                last = self._to_state(ens.get_state())
                yield dmc_base.SamplingBlock(props, iter_density, iter_ssf,
                                             last)
                block_idx += 1
        finally:
            ens.close()
            eng.close()

    def state_data_blocks(self, ini_state: State, num_time_steps_block: int
                          ) -> t.Iterator[dmc_base.SamplingStateDataBlock]:
        """qmc_base/dmc.py:988-1068; note the reference overrides the target
        population with the initial one here (:1008)."""
        nts = int(num_time_steps_block)
        eng, ens = self._start(ini_state, target=int(ini_state.num_walkers))
        maxw = self.max_num_walkers
        try:
            while True:
                confs = np.zeros((nts,) + self.state_confs_shape)
                en, wt = np.zeros((nts, maxw)), np.zeros((nts, maxw))
                mk = np.zeros((nts, maxw), dtype=bool)
                rows = []
                for k in range(nts):
                    ser = ens.run_block(1)
                    s = ens.get_state()
                    confs[k], en[k], wt[k], mk[k] = (s.confs, s.energy,
                                                     s.weight, s.mask)
                    rows.append(ser)
                props = dmc_base.PropsData(
                    *[np.concatenate([getattr(r, f) for r in rows])
                      for f in dmc_base.PropsData._fields])
                yield dmc_base.SamplingStateDataBlock(
                    confs, dmc_base.StateProps(en, wt, mk), props)
        finally:
            ens.close()
            eng.close()

    @property
    def ssf_momenta(self):
        if self.ssf_est_spec is None:
            raise TypeError('the static structure factor spec has no been '
                            'specified')
        return (np.arange(self.ssf_est_spec.num_modes) * 2 * pi /
                self.model_spec.supercell_size)
