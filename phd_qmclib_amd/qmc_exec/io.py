"""HDF5 result files (reference: qmc_exec/io.py:24-248, qmc_exec/vmc/io.py,
qmc_exec/dmc/io.py, mrbp_qmc/{vmc,dmc}_exec/io.py).

File layout, identical to the reference's so existing analysis code reads our
output:

    <group>/{vmc|dmc}/state       datasets + attributes of the last State
    <group>/{vmc|dmc}/proc_spec   attributes = Proc.as_config(); sub-groups
                                  model_spec, ssf_spec, density_spec
    <group>/{vmc|dmc}/data/blocks/<property>/totals[, weight_totals]

Access goes through `util.h5lite.open_file`: h5py when it is importable,
otherwise the ctypes facade over libhdf5.
"""
import pathlib
import typing as t

import attr
import numpy as np

from ..util import h5lite

__all__ = ['HDF5FileHandler', 'HDF5FileHandlerGroupError',
           'IO_FILE_HANDLER_TYPES', 'IO_HANDLER_TYPES']

IO_HANDLER_TYPES = ('HDF5_FILE',)
IO_FILE_HANDLER_TYPES = ('HDF5_FILE',)


class HDF5FileHandlerGroupError(ValueError):
    """The group to write to already exists (and dump_replace is off)."""


def _plain(value):
    """numpy scalars / 1-element arrays of attributes -> Python values."""
    if isinstance(value, np.ndarray) and value.shape == ():
        value = value[()]
    if isinstance(value, np.generic):
        return value.item()
    if isinstance(value, bytes):
        return value.decode('utf-8')
    return value


def attrs_dict(group) -> dict:
    return {k: _plain(v) for k, v in group.attrs.items()}


@attr.s(auto_attribs=True, frozen=True)
class HDF5FileHandler:
    """A handler for properly structured HDF5 files."""

    #: Path to the file.
    location: str = attr.ib(
        validator=attr.validators.instance_of((str, pathlib.Path)))

    #: The HDF5 group in the file to read and/or write data.
    group: str = attr.ib(validator=attr.validators.instance_of(str))

    #: Replace any existing data in the file.
    dump_replace: bool = attr.ib(
        default=False, validator=attr.validators.instance_of(bool))

    #: A tag to identify this handler.
    type: t.Optional[str] = attr.ib(default=None)

    #: 'vmc' or 'dmc' (fixed by the subclass).
    sampling_type: t.ClassVar[str] = ''

    def __attrs_post_init__(self):
        object.__setattr__(self, 'type', 'HDF5_FILE')
        if isinstance(self.location, pathlib.Path):
            object.__setattr__(self, 'location', str(self.location))
        if self.location_path.is_dir():
            raise ValueError(f"location {self.location_path} is a directory, "
                             f"not a file")

    @classmethod
    def from_config(cls, config: t.Mapping):
        return cls(**dict(config))

    @property
    def location_path(self):
        return pathlib.Path(self.location).absolute()

    # -- to be provided by the concrete handlers ------------------------
    def save_state(self, state, group):
        raise NotImplementedError

    def load_state(self, group):
        raise NotImplementedError

    def build_proc(self, proc_config: t.Dict):
        raise NotImplementedError

    def build_result(self, state, proc_inst, sampling_data):
        raise NotImplementedError

    def load_sampling_data(self, group):
        raise NotImplementedError

    # -- reference: qmc_exec/io.py:76-132 -------------------------------
    def load(self):
        """Load a procedure result from the file."""
        h5_file = h5lite.open_file(self.location_path, 'r')
        with h5_file:
            qmc_group = h5_file.get(f'{self.group}/{self.sampling_type}')
            if qmc_group is None:
                raise KeyError(f"no group '{self.group}/{self.sampling_type}'"
                               f" in {self.location_path}")
            state = self.load_state(qmc_group.get('state'))
            proc_inst = self.load_proc(qmc_group.get('proc_spec'))
            sampling_data = self.load_sampling_data(qmc_group.get('data'))
        return self.build_result(state, proc_inst, sampling_data)

    def dump(self, proc_result):
        """Save a procedure result to the file."""
        self.location_path.parent.mkdir(parents=True, exist_ok=True)
        h5_file = h5lite.open_file(self.location_path, 'a')
        with h5_file:
            base_group = h5_file.require_group(self.group)
            sampling_type = self.sampling_type
            if sampling_type in base_group:
                if self.dump_replace:
                    del base_group[sampling_type]
                else:
                    raise HDF5FileHandlerGroupError(
                        f"Unable to create '{sampling_type}' group (name "
                        f"already exists)")
            qmc_group = base_group.require_group(sampling_type)
            self.save_state(proc_result.state,
                            qmc_group.require_group('state'))
            self.save_proc(proc_result.proc.as_config(),
                           qmc_group.require_group('proc_spec'))
            proc_result.data.hdf5_export(qmc_group.require_group('data'))
            h5_file.flush()

    # -- reference: qmc_exec/io.py:157-212 ------------------------------
    def load_proc(self, group):
        proc_config = {'model_spec': attrs_dict(group.get('model_spec'))}
        for name in ('density_spec', 'ssf_spec'):
            sub = group.get(name)
            proc_config[name] = None if sub is None else attrs_dict(sub)
        proc_config.update(attrs_dict(group))
        return self.build_proc(proc_config)

    @staticmethod
    def save_proc(config: t.Dict, group):
        config = dict(config)
        group.require_group('model_spec').attrs.update(
            **config.pop('model_spec'))
        for name in ('density_spec', 'ssf_spec'):
            sub = config.pop(name, None)
            if sub is not None:
                group.require_group(name).attrs.update(**sub)
        group.attrs.update(config)
