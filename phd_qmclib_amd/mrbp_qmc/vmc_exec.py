"""VMC procedure of the Bloch-Phonon model (reference:
mrbp_qmc/vmc_exec/proc.py:155-299): same fields and defaults."""
import functools
import typing as t
import warnings

import attr
import numpy as np

from ..qmc_base import vmc as vmc_base
from ..qmc_exec import proc as proc_base
from . import model, vmc

__all__ = ['ModelSysConfSpec', 'Proc', 'ProcInput', 'ProcResult',
           'SSFEstSpec']

MODEL_SYS_CONF_TYPE = 'MODEL_SYS_CONF'
ProcInputError = proc_base.ProcInputError


def _as_int(v):
    return int(v) if isinstance(v, (int, np.integer)) and \
        not isinstance(v, bool) else v


def _opt(f):
    return lambda v: None if v is None else f(v)


@attr.s(auto_attribs=True, frozen=True)
class ModelSysConfSpec:
    """How to build the initial configuration (vmc_exec/proc.py:27-69)."""
    dist_type: str = attr.ib(validator=attr.validators.instance_of(str))
    num_sys_conf: t.Optional[int] = None
    type: str = attr.ib(default=None)

    def __attrs_post_init__(self):
        object.__setattr__(self, 'type', MODEL_SYS_CONF_TYPE)

    @classmethod
    def from_config(cls, config: t.Mapping):
        return cls(**dict(config))

    def dist_type_as_type(self):
        if self.dist_type is None:
            return model.SysConfDistType.RANDOM
        if self.dist_type not in model.SysConfDistType.__members__:
            raise ValueError
        return model.SysConfDistType[self.dist_type]


@attr.s(auto_attribs=True, frozen=True)
class SSFEstSpec:
    num_modes: int = attr.ib(converter=_as_int,
                             validator=attr.validators.instance_of(int))


@attr.s(auto_attribs=True)
class ProcInput:
    """vmc_exec/proc.py:86-123."""
    state: vmc_base.State

    @classmethod
    def from_model_sys_conf_spec(cls, sys_conf_spec: ModelSysConfSpec,
                                 proc: 'Proc'):
        sys_conf = proc.model_spec.init_get_sys_conf(
            dist_type=sys_conf_spec.dist_type_as_type())
        return cls(proc.sampling.build_state(sys_conf))

    @classmethod
    def from_result(cls, proc_result: 'ProcResult', proc: 'Proc'):
        assert proc.model_spec == proc_result.proc.model_spec
        return cls(proc_result.state)


@attr.s(auto_attribs=True, frozen=True)
class ProcResult:
    state: vmc_base.State
    proc: 'Proc'
    data: t.Any


@attr.s(auto_attribs=True, frozen=True)
class Proc:
    """VMC sampling procedure (defaults: 8 blocks x 4096 steps)."""

    model_spec: model.Spec = attr.ib(
        validator=attr.validators.instance_of(model.Spec))
    move_spread: float = attr.ib(converter=float)
    rng_seed: t.Optional[int] = attr.ib(default=None, converter=_opt(_as_int))
    num_blocks: int = attr.ib(default=8, converter=_as_int,
                              validator=attr.validators.instance_of(int))
    num_steps_block: int = attr.ib(default=4096, converter=_as_int,
                                   validator=attr.validators.instance_of(int))
    burn_in_blocks: t.Optional[int] = attr.ib(default=None,
                                              converter=_opt(_as_int))
    keep_iter_data: bool = attr.ib(default=False, converter=bool)
    density_spec: t.Optional[t.Any] = None
    ssf_spec: t.Optional[SSFEstSpec] = None

    @classmethod
    def from_config(cls, config: t.Mapping):
        """vmc_exec/proc.py:189-241, including the deprecated aliases."""
        cfg = dict(config)
        for old, new in (('num_batches', 'num_blocks'),
                         ('num_steps_batch', 'num_steps_block'),
                         ('burn_in_batches', 'burn_in_blocks')):
            if old in cfg:
                warnings.warn(f"{old} attribute is deprecated, use {new} "
                              f"instead", DeprecationWarning)
                cfg[new] = cfg.pop(old)
        model_spec = model.Spec(**cfg.pop('model_spec'))
        ssf_cfg = cfg.pop('ssf_spec', None)
        ssf = SSFEstSpec(**ssf_cfg) if ssf_cfg is not None else None
        return cls(model_spec=model_spec, ssf_spec=ssf, **cfg)

    def as_config(self):
        return attr.asdict(self, filter=attr.filters.exclude(type(None)))

    @property
    def should_eval_density(self):
        return self.density_spec is not None

    @property
    def should_eval_ssf(self):
        return self.ssf_spec is not None

    @functools.cached_property
    def sampling(self) -> vmc.Sampling:
        ssf = vmc.SSFEstSpec(self.ssf_spec.num_modes) \
            if self.should_eval_ssf else None
        return vmc.Sampling(self.model_spec, self.move_spread, self.rng_seed,
                            ssf_est_spec=ssf)

    def build_result(self, state, data):
        return ProcResult(state, self, data)

    def exec(self, proc_input: ProcInput):
        return proc_base.exec_vmc(self, proc_input)


# ---- result files + command-line application --------------------------
from ..qmc_exec import cli_app as _qcli, config as _qcfg, io as _qio  # noqa: E402
from ..qmc_exec.data import vmc as _vmc_data  # noqa: E402


@attr.s(auto_attribs=True, frozen=True)
class HDF5FileHandler(_qio.HDF5FileHandler):
    """Structured HDF5 files with VMC procedure results
    (mrbp_qmc/vmc_exec/io.py, qmc_exec/vmc/io.py:13-80)."""
    sampling_type: t.ClassVar[str] = 'vmc'

    def save_state(self, state, group):
        group.create_dataset('sys_conf', data=state.sys_conf)
        group.attrs.update({'wf_abs_log': float(state.wf_abs_log),
                            'move_stat': int(state.move_stat)})

    def load_state(self, group):
        return vmc_base.State(sys_conf=group.get('sys_conf')[()],
                              **_qio.attrs_dict(group))

    def build_proc(self, proc_config):
        return Proc.from_config({k: v for k, v in proc_config.items()
                                 if v is not None})

    def build_result(self, state, proc_inst, sampling_data):
        return ProcResult(state, proc_inst, sampling_data)

    def load_sampling_data(self, group):
        return _vmc_data.SamplingData.from_hdf5_data(group)


AppSpec, CLIApp, get_io_handler = _qcli.make_app_classes(
    Proc, ProcInput, ModelSysConfSpec, HDF5FileHandler, MODEL_SYS_CONF_TYPE)
AppMeta = _qcli.AppMeta
config_loader = _qcfg.Loader(_qio.IO_FILE_HANDLER_TYPES)
__all__ += ['AppMeta', 'AppSpec', 'CLIApp', 'HDF5FileHandler',
            'config_loader', 'get_io_handler']
