"""VMC sampling of the Bloch-Phonon model on the GPU.

`Sampling` keeps the reference surface (mrbp_qmc/vmc.py:70-171): the same
attrs fields, `build_state`, `states`, `blocks`, `as_chain`,
`state_data_blocks`, `cfc_spec`, `tpf_params`; one Markov chain, arrays shaped
like the reference's.  `EnsembleSampling` is the extension that makes the GPU
worthwhile: W independent chains (Philox stream = chain index) advanced
together, yielding per-chain block sums.

Every step runs on the device through the C-ABI (vmc_block_kernel); there is
no CPU path.
"""
import typing as t
from math import pi, sqrt

import attr
import numpy as np

from .. import utils
from ..engine import ModelEngine, VmcEnsemble
from ..qmc_base import vmc as vmc_base
from . import model

__all__ = ['CFCSpec', 'EnsembleSampling', 'Sampling', 'NDFSampling',
           'SSFEstSpec', 'SSFParams', 'StateError', 'TPFParams']

STAT_ACCEPTED = vmc_base.STAT_ACCEPTED
STAT_REJECTED = vmc_base.STAT_REJECTED


class TPFParams(t.NamedTuple):
    """Transition probability parameters (mrbp_qmc/vmc.py:29-38)."""
    boson_number: int
    move_spread: float
    lower_bound: float
    upper_bound: float


class SSFParams(t.NamedTuple):
    num_modes: int
    supercell_size: float
    assume_none: bool = False


class CFCSpec(t.NamedTuple):
    model_params: model.Params
    obf_params: model.OBFParams
    tbf_params: model.TBFParams
    tpf_params: TPFParams
    ssf_params: t.Optional[SSFParams] = None


class StateError(ValueError):
    """Flags errors related to the handling of a VMC state."""


@attr.s(auto_attribs=True)
class SSFEstSpec:
    """Structure factor estimator spec (mrbp_qmc/vmc.py:61-66)."""
    num_modes: int


def _fourier_density(momenta, pos):
    """S(k) parts of one configuration (qmc_base/jastrow/model.py:977-1002,
    jastrow/vmc.py:340-349): [num_modes, 3] = |rho_k|^2, Re rho_k, Im rho_k.
    Host-side estimator over device-generated configurations."""
    ph = momenta[:, None] * pos[None, :]
    re, im = np.cos(ph).sum(axis=1), np.sin(ph).sum(axis=1)
    return np.stack([re * re + im * im, re, im], axis=1)


@attr.s(auto_attribs=True, frozen=True)
class Sampling:
    """The spec of a (single chain) VMC sampling, uniform proposal."""

    model_spec: model.Spec
    move_spread: float
    rng_seed: t.Optional[int] = attr.ib(default=None)
    ssf_est_spec: t.Optional[SSFEstSpec] = None

    _gaussian: t.ClassVar[bool] = False

    def __attrs_post_init__(self):
        if self.rng_seed is None:
            object.__setattr__(self, 'rng_seed',
                               int(utils.get_random_rng_seed()))

    # -- parameters (mrbp_qmc/vmc.py:88-143) -----------------------------------
    @property
    def tpf_params(self):
        z_min, z_max = self.model_spec.boundaries
        return TPFParams(self.model_spec.boson_number, self.move_spread,
                         z_min, z_max)

    @property
    def ssf_params(self):
        L = self.model_spec.supercell_size
        if self.ssf_est_spec is None:
            return SSFParams(1, L, assume_none=True)
        return SSFParams(self.ssf_est_spec.num_modes, L, assume_none=False)

    @property
    def cfc_spec(self) -> CFCSpec:
        ms = self.model_spec
        return CFCSpec(ms.params, ms.obf_params, ms.tbf_params,
                       self.tpf_params, self.ssf_params)

    @property
    def ssf_momenta(self):
        if self.ssf_est_spec is None:
            raise TypeError('the static structure factor spec has no been '
                            'specified')
        return (np.arange(self.ssf_est_spec.num_modes) * 2 * pi /
                self.model_spec.supercell_size)

    @property
    def core_funcs(self):
        return model.core_funcs

    # -- engine plumbing ----------------------------------------------------------
    def _proposal_width(self):
        return self.move_spread

    def _engine(self) -> ModelEngine:
        return ModelEngine(self.model_spec.cfc_spec)

    def build_state(self, sys_conf: np.ndarray) -> vmc_base.State:
        """mrbp_qmc/vmc.py:145-165."""
        sys_conf = np.asarray(sys_conf)
        if sys_conf.shape != self.model_spec.sys_conf_shape:
            raise StateError("sys_conf is not a valid configuration "
                             "of the model spec")
        wf = model.core_funcs.wf_abs_log(sys_conf, *self.model_spec.cfc_spec)
        return vmc_base.State(sys_conf, wf, STAT_ACCEPTED)

    def set_replay_tape(self, tape):
        """TEST ONLY: the next generator replays a recorded random stream
        instead of Philox -- tape[steps, N + 1] = the N proposal draws and
        the accept uniform of every step, in the reference's call order."""
        object.__setattr__(self, '_replay_tape',
                           None if tape is None else
                           np.ascontiguousarray(tape, dtype=np.float64))

    def _start(self, ini_state: vmc_base.State):
        eng = self._engine()
        ens = VmcEnsemble(eng, 1, self._proposal_width(), self.rng_seed,
                          gaussian=self._gaussian)
        pos = np.asarray(ini_state.sys_conf, dtype=np.float64)[model.SysConfSlot.pos]
        ens.set_state(pos[None, :])
        tape = getattr(self, '_replay_tape', None)
        if tape is not None:
            ens.set_tape(tape[None, :, :])
        return eng, ens

    def _state_from(self, ens, wf, move_stat):
        pos, _, _ = ens.get_state()
        sys_conf = self.model_spec.get_sys_conf_buffer()
        sys_conf[model.SysConfSlot.pos] = pos[0]
        return vmc_base.State(sys_conf, float(wf), int(move_stat))

    # -- generators (qmc_base/vmc.py:204-251) -----------------------------------
    def states(self, ini_state: vmc_base.State) -> t.Iterator[vmc_base.State]:
        """Yields a State per step; the first one is the initial state flagged
        ACCEPTED (qmc_base/vmc.py:616-618)."""
        eng, ens = self._start(ini_state)
        try:
            while True:
                out = ens.run_block(1, sums=False, series=True)
                yield self._state_from(ens, out['wf_abs_log'][0, 0],
                                       out['move_stat'][0, 0])
        finally:
            ens.close()
            eng.close()

    def blocks(self, num_steps_block: int, ini_state: vmc_base.State
               ) -> t.Iterator[vmc_base.SamplingBlock]:
        """qmc_base/vmc.py:686-768: per block the series wf_abs_log, energy,
        move_stat, the S(k) parts when requested, accept_rate, last_state."""
        ns = int(num_steps_block)
        eng, ens = self._start(ini_state)
        want_ssf = self.ssf_est_spec is not None
        momenta = self.ssf_momenta if want_ssf else None
        try:
            while True:
                out = ens.run_block(ns, series=True, confs=want_ssf)
                props = vmc_base.PropsData(out['wf_abs_log'][:, 0].copy(),
                                           out['energy'][:, 0].copy(),
                                           out['move_stat'][:, 0].copy())
                iter_ssf = None
                if want_ssf:
                    iter_ssf = np.stack([_fourier_density(momenta, p[0])
                                         for p in out['pos']])
                last = self._state_from(ens, props.wf_abs_log[-1],
                                        props.move_stat[-1])
                yield vmc_base.SamplingBlock(
                    props, iter_ssf, float(out['num_accepted'][0]) / ns, last)
        finally:
            ens.close()
            eng.close()

    def state_data_blocks(self, num_steps_block: int,
                          ini_state: vmc_base.State):
        """qmc_base/vmc.py:825-900: blocks that keep every configuration."""
        ns = int(num_steps_block)
        n = self.model_spec.boson_number
        eng, ens = self._start(ini_state)
        try:
            while True:
                out = ens.run_block(ns, series=True, confs=True)
                confs = np.zeros((ns,) + self.model_spec.sys_conf_shape)
                confs[:, model.SysConfSlot.pos, :] = out['pos'][:, 0, :n]
                props = vmc_base.PropsData(out['wf_abs_log'][:, 0].copy(),
                                           out['energy'][:, 0].copy(),
                                           out['move_stat'][:, 0].copy())
                last = vmc_base.State(confs[-1].copy(),
                                      float(props.wf_abs_log[-1]),
                                      int(props.move_stat[-1]))
                yield vmc_base.SamplingStateDataBlock(
                    confs, props, float(out['num_accepted'][0]) / ns, last)
        finally:
            ens.close()
            eng.close()

    def as_chain(self, num_steps: int, ini_state: vmc_base.State):
        """qmc_base/vmc.py:215-229."""
        if not num_steps >= 1:
            raise ValueError('num_steps must be nonzero and positive')
        gen = self.state_data_blocks(num_steps, ini_state)
        try:
            return next(gen)
        finally:
            gen.close()


@attr.s(auto_attribs=True, frozen=True)
class NDFSampling(Sampling):
    """Gaussian-proposal VMC (mrbp_qmc/vmc_ndf.py:23-51): `time_step` is the
    variance of the normal displacement."""

    model_spec: model.Spec
    time_step: float = None
    rng_seed: t.Optional[int] = attr.ib(default=None)
    ssf_est_spec: t.Optional[SSFEstSpec] = None
    move_spread: float = attr.ib(default=None, init=False)

    _gaussian: t.ClassVar[bool] = True

    def _proposal_width(self):
        return sqrt(self.time_step)


@attr.s(auto_attribs=True)
class EnsembleSampling:
    """W independent VMC chains advanced together on one GPU (extension of the
    reference's single-chain sampling: every chain is statistically the chain
    `Sampling` generates, with Philox stream = global chain index)."""

    model_spec: model.Spec
    move_spread: float
    num_chains: int
    rng_seed: t.Optional[int] = None
    first_chain: int = 0          # global index of chain 0 (multi-GPU shard)
    device: t.Optional[int] = None
    stream: t.Optional[int] = None

    def __attrs_post_init__(self):
        if self.rng_seed is None:
            self.rng_seed = int(utils.get_random_rng_seed())
        self.engine = ModelEngine(self.model_spec.cfc_spec, device=self.device,
                                  stream=self.stream)
        self.ensemble = VmcEnsemble(self.engine, self.num_chains,
                                    self.move_spread, self.rng_seed,
                                    chain0=self.first_chain)

    def set_confs(self, pos):
        """pos[W, N] initial positions (log|psi| is evaluated on the device)."""
        self.ensemble.set_state(pos)

    def init_random(self, seed=None):
        rng = np.random.RandomState(seed)
        L = self.model_spec.supercell_size
        self.set_confs(L * rng.random_sample((self.num_chains,
                                               self.model_spec.boson_number)))

    def blocks(self, num_steps_block: int) -> t.Iterator[vmc_base.EnsembleBlock]:
        ns = int(num_steps_block)
        while True:
            out = self.ensemble.run_block(ns)
            yield vmc_base.EnsembleBlock(out['sum_energy'], out['sum_energy2'],
                                         out['num_accepted'], ns)

    def confs(self):
        """Current positions of every chain, [W, N] (VMC -> DMC hand-off)."""
        return self.ensemble.get_state()[0]

    def ssf(self, num_modes: int):
        """Static structure factor of the ensemble's current configurations,
        S(k_m) = (<|rho_k|^2> - <Re rho_k>^2 - <Im rho_k>^2) / N with the
        averages over the chains (qmc_exec/data/vmc.py SSFBlocks.mean), and
        the momenta k_m = 2 pi m / L -> (momenta[M], ssf[M])."""
        parts = self.ensemble.ssf_parts(num_modes)
        n, L = self.model_spec.boson_number, self.model_spec.supercell_size
        ssf = (parts[:, 0] - parts[:, 1] ** 2 - parts[:, 2] ** 2) / n
        return np.arange(int(num_modes)) * 2 * pi / L, ssf

    def close(self):
        self.ensemble.close()
        self.engine.close()
