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
