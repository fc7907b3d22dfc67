"""VMC block containers (reference: qmc_exec/data/vmc.py:23-122)."""
import typing as t

import attr
import numpy as np

from ...stats import reblock

__all__ = ['EnergyBlocks', 'PropBlocks', 'PropsDataBlocks', 'SamplingData']


@attr.s(auto_attribs=True, frozen=True)
class PropBlocks:
    """A series of block values; mean and error by reblocking."""
    totals: np.ndarray

    @property
    def reblock(self):
        return reblock.OTFObject.from_non_obj_data(self.totals)

    @property
    def mean(self):
        return self.reblock.mean

    @property
    def mean_error(self):
        return self.reblock.mean_eff_error

    def __len__(self):
        return len(self.totals)

    def __add__(self, other):
        if not isinstance(other, PropBlocks):
            return NotImplemented
        return type(self)(np.concatenate((self.totals, other.totals), axis=0))


@attr.s(auto_attribs=True, frozen=True)
class EnergyBlocks(PropBlocks):
    totals: np.ndarray

    @classmethod
    def from_data(cls, data, reduce_data: bool = True):
        """Block value = mean of the block's energy series
        (qmc_exec/data/vmc.py:108-122)."""
        energy = np.asarray(data.energy)
        return cls(energy.mean(axis=1) if reduce_data else energy)


@attr.s(auto_attribs=True, frozen=True)
class PropsDataBlocks:
    energy: EnergyBlocks
    ss_factor: t.Optional[t.Any] = None


@attr.s(auto_attribs=True, frozen=True)
class SamplingData:
    blocks: PropsDataBlocks
    series: t.Optional[t.Any] = None


# ---- HDF5 layout (qmc_exec/data/vmc.py:44-62, 246-276, 336-373, 408-429):
# blocks/energy/totals; blocks/ss_factor/{fdk_sqr_abs,fdk_real,fdk_imag}/totals
_SSF_PARTS = ('fdk_sqr_abs', 'fdk_real', 'fdk_imag')


def _prop_export(self, group):
    group.create_dataset('totals', data=self.totals)


def _prop_import(cls, group):
    return cls(totals=group.get('totals')[()])


PropBlocks.hdf5_export = _prop_export
PropBlocks.from_hdf5_data = classmethod(_prop_import)


def _blocks_export(self, group):
    self.energy.hdf5_export(group.require_group('energy'))
    if self.ss_factor is not None:
        ssf_group = group.require_group('ss_factor')
        tot = np.asarray(self.ss_factor.totals)      # [block, mode, part]
        for c, name in enumerate(_SSF_PARTS):
            ssf_group.require_group(name).create_dataset(
                'totals', data=np.ascontiguousarray(tot[..., c]))


def _blocks_import(cls, group):
    energy = EnergyBlocks.from_hdf5_data(group.get('energy'))
    ssf_group, ssf = group.get('ss_factor'), None
    if ssf_group is not None:
        ssf = PropBlocks(np.stack([ssf_group.get(name).get('totals')[()]
                                   for name in _SSF_PARTS], axis=-1))
    return cls(energy, ssf)


PropsDataBlocks.hdf5_export = _blocks_export
PropsDataBlocks.from_hdf5_data = classmethod(_blocks_import)


def _sampling_export(self, group):
    self.blocks.hdf5_export(group.require_group('blocks'))


def _sampling_import(cls, group):
    return cls(PropsDataBlocks.from_hdf5_data(group.get('blocks')))


SamplingData.hdf5_export = _sampling_export
SamplingData.from_hdf5_data = classmethod(_sampling_import)
