"""VMC block containers (reference: qmc_exec/data/vmc.py:23-122)."""
import typing as t

import attr
import numpy as np

from ...stats import reblock

__all__ = ['DensityBlocks', 'EnergyBlocks', 'PropBlocks', 'PropsDataBlocks',
           'SSFBlocks', 'SSFPartBlocks', 'SamplingData']


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
class DensityBlocks(PropBlocks):
    """Density data in blocks (qmc_exec/data/vmc.py:125-144)."""
    totals: np.ndarray

    @classmethod
    def from_data(cls, density_data, reduce_data: bool = True):
        density_data = np.asarray(density_data)
        return cls(density_data.mean(axis=1) if reduce_data else density_data)


@attr.s(auto_attribs=True, frozen=True)
class SSFPartBlocks(PropBlocks):
    """One part (|rho_k|^2, Re rho_k or Im rho_k) of the structure factor in
    blocks, one column per momentum: reblocked as a set
    (qmc_exec/data/vmc.py:147-171)."""
    totals: np.ndarray

    @classmethod
    def from_data(cls, ssf_data, reduce_data: bool = True):
        ssf_data = np.asarray(ssf_data)
        return cls(ssf_data.mean(axis=1) if reduce_data else ssf_data)

    @property
    def reblock(self):
        return reblock.OTFSet.from_non_obj_data(self.totals)


@attr.s(auto_attribs=True, frozen=True)
class SSFBlocks:
    """Static structure factor in blocks (qmc_exec/data/vmc.py:174-262):
    S(k) = <|rho_k|^2> - <Re rho_k>^2 - <Im rho_k>^2."""
    fdk_sqr_abs_part: SSFPartBlocks
    fdk_real_part: SSFPartBlocks
    fdk_imag_part: SSFPartBlocks

    @classmethod
    def from_data(cls, ssf_data, reduce_data: bool = True):
        """ssf_data[block, (step,) mode, 3]; with `reduce_data` the block
        value is the mean over the steps of the block."""
        ssf_data = np.asarray(ssf_data)
        totals = ssf_data.mean(axis=1) if reduce_data else ssf_data
        return cls(SSFPartBlocks(totals[:, :, 0]), SSFPartBlocks(totals[:, :, 1]),
                   SSFPartBlocks(totals[:, :, 2]))

    @property
    def mean(self):
        return (self.fdk_sqr_abs_part.mean - self.fdk_real_part.mean ** 2 -
                self.fdk_imag_part.mean ** 2)

    @property
    def mean_error(self):
        re, im = self.fdk_real_part, self.fdk_imag_part
        return (self.fdk_sqr_abs_part.mean_error +
                2 * (re.mean * re.mean_error + im.mean * im.mean_error))

    def __len__(self):
        return len(self.fdk_sqr_abs_part)

    def __add__(self, other):
        if not isinstance(other, SSFBlocks):
            return NotImplemented
        return SSFBlocks(self.fdk_sqr_abs_part + other.fdk_sqr_abs_part,
                         self.fdk_real_part + other.fdk_real_part,
                         self.fdk_imag_part + other.fdk_imag_part)


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


def _ssf_export(self, group):
    for name, part in zip(_SSF_PARTS, (self.fdk_sqr_abs_part,
                                       self.fdk_real_part,
                                       self.fdk_imag_part)):
        part.hdf5_export(group.require_group(name))


def _ssf_import(cls, group):
    return cls(*[SSFPartBlocks.from_hdf5_data(group.get(name))
                 for name in _SSF_PARTS])


SSFBlocks.hdf5_export = _ssf_export
SSFBlocks.from_hdf5_data = classmethod(_ssf_import)


def _blocks_export(self, group):
    self.energy.hdf5_export(group.require_group('energy'))
    if self.ss_factor is not None:
        self.ss_factor.hdf5_export(group.require_group('ss_factor'))


def _blocks_import(cls, group):
    energy = EnergyBlocks.from_hdf5_data(group.get('energy'))
    ssf_group, ssf = group.get('ss_factor'), None
    if ssf_group is not None:
        ssf = SSFBlocks.from_hdf5_data(ssf_group)
    return cls(energy, ssf)


PropsDataBlocks.hdf5_export = _blocks_export
PropsDataBlocks.from_hdf5_data = classmethod(_blocks_import)


def _sampling_export(self, group):
    self.blocks.hdf5_export(group.require_group('blocks'))


def _sampling_import(cls, group):
    return cls(PropsDataBlocks.from_hdf5_data(group.get('blocks')))


SamplingData.hdf5_export = _sampling_export
SamplingData.from_hdf5_data = classmethod(_sampling_import)
