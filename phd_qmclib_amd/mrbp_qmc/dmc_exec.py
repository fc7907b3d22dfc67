"""DMC procedure of the Bloch-Phonon model (reference:
mrbp_qmc/dmc_exec/proc.py:161-398): same fields and defaults (512 blocks x 512
time steps, 480 target / 512 max walkers, control factor 0.5)."""
import functools
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

    @functools.cached_property
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


# ---- result files + command-line application --------------------------
from ..qmc_exec import cli_app as _qcli, config as _qcfg, io as _qio  # noqa: E402
from ..qmc_exec.data import dmc as _dmc_data  # noqa: E402
from .vmc_exec import MODEL_SYS_CONF_TYPE  # noqa: E402

#: on-disk record of the cloning table (qmc_base/dmc.py:381-384)
branching_spec_dtype = np.dtype([('CLONING_FACTOR', np.int32),
                                 ('CLONING_REF', np.int32)])


@attr.s(auto_attribs=True, frozen=True)
class HDF5FileHandler(_qio.HDF5FileHandler):
    """Structured HDF5 files with DMC procedure results
    (mrbp_qmc/dmc_exec/io.py, qmc_exec/dmc/io.py:11-98)."""
    sampling_type: t.ClassVar[str] = 'dmc'

    def save_state(self, state, group):
        maxw = int(state.max_num_walkers)
        rec = np.zeros(maxw, dtype=branching_spec_dtype)
        bs = state.branching_spec
        if bs is not None:
            if bs.cloning_factor is not None:
                rec['CLONING_FACTOR'] = np.asarray(bs.cloning_factor)[:maxw]
            rec['CLONING_REF'] = np.asarray(bs.cloning_ref)[:maxw]
        group.create_dataset('branching_spec', data=rec)
        group.create_dataset('confs', data=state.confs)
        props_group = group.require_group('props')
        props_group.create_dataset('energy', data=state.props.energy)
        props_group.create_dataset('weight', data=state.props.weight)
        props_group.create_dataset('mask',
                                   data=np.asarray(state.props.mask, bool))
        group.attrs.update({
            'energy': float(state.energy), 'weight': float(state.weight),
            'num_walkers': int(state.num_walkers),
            'ref_energy': float(state.ref_energy),
            'accum_energy': float(state.accum_energy),
            'max_num_walkers': maxw})

    def load_state(self, group):
        rec = group.get('branching_spec')[()]
        props = group.get('props')
        state_props = dmc_base.StateProps(props.get('energy')[()],
                                          props.get('weight')[()],
                                          props.get('mask')[()])
        bs = dmc_base.BranchingSpec(
            np.asarray(rec['CLONING_FACTOR'], dtype=np.int64),
            np.asarray(rec['CLONING_REF'], dtype=np.int64))
        return dmc_base.State(confs=group.get('confs')[()], props=state_props,
                              branching_spec=bs, **_qio.attrs_dict(group))

    def build_proc(self, proc_config):
        return Proc.from_config({k: v for k, v in proc_config.items()
                                 if v is not None})

    def build_result(self, state, proc_inst, sampling_data):
        return ProcResult(state, proc_inst, sampling_data)

    def load_sampling_data(self, group):
        return _dmc_data.SamplingData.from_hdf5_data(group)


AppSpec, CLIApp, get_io_handler = _qcli.make_app_classes(
    Proc, ProcInput, ModelSysConfSpec, HDF5FileHandler, MODEL_SYS_CONF_TYPE)
AppMeta = _qcli.AppMeta
config_loader = _qcfg.Loader(_qio.IO_FILE_HANDLER_TYPES)
__all__ += ['AppMeta', 'AppSpec', 'CLIApp', 'HDF5FileHandler',
            'config_loader', 'get_io_handler']
