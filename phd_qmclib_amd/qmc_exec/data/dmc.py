"""DMC block containers (reference: qmc_exec/data/dmc.py:24-318).

A DMC expectation value is a ratio of block totals, <O w> / <w>; its error
combines the reblocked variances of numerator, denominator and their product
(the reference's formula, restated in `PropBlocks.mean_error`).
"""
import typing as t

import attr
import numpy as np

from ...stats import reblock

__all__ = ['EnergyBlocks', 'NumWalkersBlocks', 'PropBlocks',
           'PropsDataBlocks', 'SamplingData', 'UnWeightedPropBlocks',
           'WeightBlocks']


@attr.s(auto_attribs=True, frozen=True)
class PropBlocks:
    """Weighted block totals."""
    totals: np.ndarray
    weight_totals: t.Optional[np.ndarray]

    @property
    def reblock(self):
        return reblock.OTFObject.from_non_obj_data(self.totals)

    @property
    def weight_reblock(self):
        if self.weight_totals is None:
            return None
        return reblock.OTFObject.from_non_obj_data(self.weight_totals)

    @property
    def cross_weight_reblock(self):
        if self.weight_totals is None:
            return None
        return reblock.OTFObject.from_non_obj_data(self.totals *
                                                   self.weight_totals)

    @property
    def mean(self):
        w = self.weight_reblock
        return self.reblock.mean if w is None else self.reblock.mean / w.mean

    @property
    def mean_error(self):
        """qmc_exec/data/dmc.py:41-75."""
        ow = self.reblock
        ow_mean, ow_var, ow_eff = ow.mean, ow.var, ow.eff_size
        if self.weight_reblock is None:
            w_mean, w_var, oww_mean = 1., 0., ow_mean
            w_eff = oww_eff = 0.5
        else:
            w, oww = self.weight_reblock, self.cross_weight_reblock
            w_mean, w_var, oww_mean = w.mean, w.var, oww.mean
            w_eff, oww_eff = w.eff_size, oww.eff_size
        err_ow = ow_var / ow_mean ** 2
        err_w = w_var / w_mean ** 2
        err_oww = (oww_mean - ow_mean * w_mean) / (ow_mean * w_mean)
        return np.abs(self.mean) * np.sqrt(err_ow / ow_eff + err_w / w_eff -
                                           2 * err_oww / oww_eff)

    def __len__(self):
        return len(self.totals)

    def __add__(self, other):
        if not isinstance(other, PropBlocks):
            return NotImplemented
        return type(self)(
            np.concatenate((self.totals, other.totals), axis=0),
            np.concatenate((self.weight_totals, other.weight_totals), axis=0))


@attr.s(auto_attribs=True, frozen=True)
class UnWeightedPropBlocks:
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
        if not isinstance(other, UnWeightedPropBlocks):
            return NotImplemented
        return type(self)(np.concatenate((self.totals, other.totals), axis=0))


@attr.s(auto_attribs=True, frozen=True)
class NumWalkersBlocks(UnWeightedPropBlocks):
    totals: np.ndarray

    @classmethod
    def from_data(cls, data, reduce_data: bool = True):
        nw = np.asarray(data.num_walkers)
        return cls(nw.sum(axis=1) if reduce_data else nw)


@attr.s(auto_attribs=True, frozen=True)
class WeightBlocks(UnWeightedPropBlocks):
    totals: np.ndarray

    @classmethod
    def from_data(cls, data, reduce_data: bool = True):
        w = np.asarray(data.weight)
        return cls(w.sum(axis=1) if reduce_data else w)


@attr.s(auto_attribs=True, frozen=True)
class EnergyBlocks(PropBlocks):
    totals: np.ndarray
    weight_totals: np.ndarray

    @classmethod
    def from_data(cls, data, reduce_data: bool = True):
        e, w = np.asarray(data.energy), np.asarray(data.weight)
        if reduce_data:
            return cls(e.sum(axis=1), w.sum(axis=1))
        return cls(e, w)


@attr.s(auto_attribs=True, frozen=True)
class PropsDataBlocks:
    energy: EnergyBlocks
    weight: WeightBlocks
    num_walkers: NumWalkersBlocks
    density: t.Optional[t.Any] = None
    ss_factor: t.Optional[t.Any] = None


@attr.s(auto_attribs=True, frozen=True)
class SamplingData:
    blocks: PropsDataBlocks
    series: t.Optional[t.Any] = None
