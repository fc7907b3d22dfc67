"""DMC procedure of the Bloch-Phonon model (reference:
mrbp_qmc/dmc_exec/proc.py:161-398): same fields and defaults (512 blocks x 512
time steps, 480 target / 512 max walkers, control factor 0.5)."""
import typing as t
import warnings

import attr
import numpy as np

from ..qmc_base import dmc as dmc_base
from ..qmc_exec import proc as proc_base
from . import dmc, model
from .vmc_exec import ModelSysConfSpec, _as_int, _opt

__all__ = ['DensityEstSpec', 'ModelSysConfSpec', 'Proc', 'ProcInput',
           'ProcResult', 'SSFEstSpec']

ProcInputError = proc_base.ProcInputError


@attr.s(auto_attribs=True, frozen=True)
class DensityEstSpec:
    """dmc_exec/proc.py:71-81."""
    num_bins: int = attr.ib(converter=_as_int,
                            validator=attr.validators.instance_of(int))
    as_pure_est: bool = attr.ib(default=True, converter=bool)


@attr.s(auto_attribs=True, frozen=True)
class SSFEstSpec:
    """dmc_exec/proc.py:84-94."""
    num_modes: int = attr.ib(converter=_as_int,
                             validator=attr.validators.instance_of(int))
    as_pure_est: bool = attr.ib(default=True, converter=bool)


@attr.s(auto_attribs=True)
class ProcInput:
    """dmc_exec/proc.py:100-143."""
    state: dmc_base.State

    @classmethod
    def from_model_sys_conf_spec(cls, sys_conf_spec: ModelSysConfSpec,
                                 proc: 'Proc'):
        dist_type = sys_conf_spec.dist_type_as_type()
        num = sys_conf_spec.num_sys_conf or proc.target_num_walkers
        confs = np.asarray([proc.model_spec.init_get_sys_conf(
            dist_type=dist_type) for _ in range(num)])
        return cls(proc.sampling.build_state(confs))

    @classmethod
    def from_result(cls, proc_result: 'ProcResult', proc: 'Proc'):
        return cls(proc_result.state)


@attr.s(auto_attribs=True, frozen=True)
class ProcResult:
    state: dmc_base.State
    proc: 'Proc'
    data: t.Any


@attr.s(auto_attribs=True, frozen=True)
class Proc:
    """DMC sampling procedure."""

    model_spec: model.Spec = attr.ib(
        validator=attr.validators.instance_of(model.Spec))
    time_step: float = attr.ib(converter=float)
    max_num_walkers: int = attr.ib(default=512, converter=_as_int,
                                   validator=attr.validators.instance_of(int))
    target_num_walkers: int = attr.ib(default=480, converter=_as_int,
                                      validator=attr.validators.instance_of(int))
    num_walkers_control_factor: t.Optional[float] = attr.ib(default=0.5,
                                                            converter=float)
    rng_seed: t.Optional[int] = attr.ib(default=None, converter=_opt(_as_int))
    num_blocks: int = attr.ib(default=512, converter=_as_int,
                              validator=attr.validators.instance_of(int))
    num_time_steps_block: int = attr.ib(default=512, converter=_as_int,
                                        validator=attr.validators.instance_of(int))
    burn_in_blocks: t.Optional[int] = attr.ib(default=None,
                                              converter=_opt(_as_int))
    keep_iter_data: bool = attr.ib(default=False, converter=bool)
    density_spec: t.Optional[t.Any] = None
    ssf_spec: t.Optional[t.Any] = None
    jit_parallel: bool = attr.ib(default=True, converter=bool)
    jit_fastmath: bool = attr.ib(default=False, converter=bool)
    verbose: bool = attr.ib(default=False, converter=bool)

    @classmethod
    def from_config(cls, config: t.Mapping):
        """dmc_exec/proc.py:223-293, including the deprecated aliases."""
        cfg = dict(config)
        for old, new in (('num_batches', 'num_blocks'),
                         ('num_time_steps_batch', 'num_time_steps_block'),
                         ('burn_in_batches', 'burn_in_blocks')):
            if old in cfg:
                warnings.warn(f"{old} attribute is deprecated, use {new} "
                              f"instead", DeprecationWarning)
                cfg[new] = cfg.pop(old)
        for old, new in (('parallel', 'jit_parallel'),
                         ('fastmath', 'jit_fastmath')):
            if old in cfg:
                cfg[new] = cfg.pop(old)
        model_spec = model.Spec(**cfg.pop('model_spec'))
        dens_cfg = cfg.pop('density_spec', None)
        dens = DensityEstSpec(**dens_cfg) if dens_cfg is not None else None
        ssf_cfg = cfg.pop('ssf_spec', None)
        ssf = None
        if ssf_cfg is not None:
            ssf_cfg = dict(ssf_cfg)
            ssf_cfg.pop('pfw_num_time_steps', None)
            ssf = SSFEstSpec(**ssf_cfg)
        return cls(model_spec=model_spec, density_spec=dens, ssf_spec=ssf,
                   **cfg)

    def as_config(self):
        return attr.asdict(self, filter=attr.filters.exclude(type(None)))

    @property
    def should_eval_density(self):
        return self.density_spec is not None

    @property
    def should_eval_ssf(self):
        return self.ssf_spec is not None

    @property
    def sampling(self) -> dmc.Sampling:
        """dmc_exec/proc.py:336-371: the forward walking of the pure
        estimators spans one block."""
        pfw = self.num_time_steps_block
        dens = ssf = None
        if self.should_eval_density:
            dens = dmc.DensityEstSpec(self.density_spec.num_bins,
                                      self.density_spec.as_pure_est, pfw)
        if self.should_eval_ssf:
            ssf = dmc.SSFEstSpec(self.ssf_spec.num_modes,
                                 self.ssf_spec.as_pure_est, pfw)
        return dmc.Sampling(self.model_spec, self.time_step,
                            self.max_num_walkers, self.target_num_walkers,
                            self.num_walkers_control_factor, self.rng_seed,
                            density_est_spec=dens, ssf_est_spec=ssf)

    def build_result(self, state, data):
        return ProcResult(state, self, data)

    def exec(self, proc_input: ProcInput):
        return proc_base.exec_dmc(self, proc_input)
