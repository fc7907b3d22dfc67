"""DMC block containers (reference: qmc_exec/data/dmc.py:24-318).

A DMC expectation value is a ratio of block totals, <O w> / <w>; its error
combines the reblocked variances of numerator, denominator and their product
(the reference's formula, restated in `PropBlocks.mean_error`).
"""
import typing as t

import attr
import numpy as np

from ...stats import reblock

__all__ = ['DensityBlocks', 'EnergyBlocks', 'NumWalkersBlocks', 'PropBlocks',
           'SSFBlocks', 'SSFPartBlocks',
           'PropsDataBlocks', 'PropsDataSeries', 'SamplingData', 'UnWeightedPropBlocks',
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


def _est_totals(nts_block, data, weight, reduce_data, as_pure_est,
                pure_est_reduce_factor):
    """Block totals of an estimator (qmc_exec/data/dmc.py:329-371, 425-466):
    mixed estimators sum over the block, pure ones take the last time step."""
    data, weight = np.asarray(data), np.asarray(weight)
    if not as_pure_est:
        if reduce_data:
            return data.sum(axis=1), weight.sum(axis=1)[:, np.newaxis]
        return data, weight[:, np.newaxis]
    if reduce_data:
        return (data[:, nts_block - 1, :],
                weight[:, nts_block - 1][:, np.newaxis])
    return data, (weight * pure_est_reduce_factor)[:, np.newaxis]


@attr.s(auto_attribs=True, frozen=True)
class SetPropBlocks(PropBlocks):
    """Weighted block totals of a vector-valued property (one column per bin
    or momentum); statistics per column through `reblock.OTFSet`."""
    totals: np.ndarray
    weight_totals: np.ndarray

    @property
    def reblock(self):
        return reblock.OTFSet.from_non_obj_data(self.totals)

    @property
    def weight_reblock(self):
        if self.weight_totals is None:
            return None
        return reblock.OTFSet.from_non_obj_data(self.weight_totals)

    @property
    def cross_weight_reblock(self):
        if self.weight_totals is None:
            return None
        return reblock.OTFSet.from_non_obj_data(self.totals *
                                                self.weight_totals)


@attr.s(auto_attribs=True, frozen=True)
class DensityBlocks(SetPropBlocks):
    """Density data in blocks (qmc_exec/data/dmc.py:321-393)."""
    totals: np.ndarray
    weight_totals: np.ndarray

    @classmethod
    def from_data(cls, num_time_steps_block, density_data, props_data,
                  reduce_data=True, as_pure_est=True,
                  pure_est_reduce_factor=None):
        return cls(*_est_totals(num_time_steps_block, density_data,
                                props_data.weight, reduce_data, as_pure_est,
                                pure_est_reduce_factor))


@attr.s(auto_attribs=True, frozen=True)
class SSFPartBlocks(SetPropBlocks):
    totals: np.ndarray
    weight_totals: np.ndarray


@attr.s(auto_attribs=True, frozen=True)
class SSFBlocks:
    """Structure factor data in blocks (qmc_exec/data/dmc.py:495-621):
    S(k) = <|rho_k|^2> - <Re rho_k>^2 - <Im rho_k>^2."""
    fdk_sqr_abs_part: SSFPartBlocks
    fdk_real_part: SSFPartBlocks
    fdk_imag_part: SSFPartBlocks

    @classmethod
    def from_data(cls, num_time_steps_block, ssf_data, props_data,
                  reduce_data=True, as_pure_est=True,
                  pure_est_reduce_factor=None):
        totals, w = _est_totals(num_time_steps_block, ssf_data,
                                props_data.weight, reduce_data, as_pure_est,
                                pure_est_reduce_factor)
        return cls(SSFPartBlocks(totals[:, :, 0], w),
                   SSFPartBlocks(totals[:, :, 1], w),
                   SSFPartBlocks(totals[:, :, 2], w))

    @property
    def mean(self):
        return (self.fdk_sqr_abs_part.mean - self.fdk_real_part.mean ** 2 -
                self.fdk_imag_part.mean ** 2)

    @property
    def mean_error(self):
        re, im = self.fdk_real_part, self.fdk_imag_part
        return (self.fdk_sqr_abs_part.mean_error +
                2 * (re.mean * re.mean_error + im.mean * im.mean_error))

    def __add__(self, other):
        if not isinstance(other, SSFBlocks):
            return NotImplemented
        return SSFBlocks(self.fdk_sqr_abs_part + other.fdk_sqr_abs_part,
                         self.fdk_real_part + other.fdk_real_part,
                         self.fdk_imag_part + other.fdk_imag_part)


@attr.s(auto_attribs=True, frozen=True)
class PropsDataBlocks:
    energy: EnergyBlocks
    weight: WeightBlocks
    num_walkers: NumWalkersBlocks
    density: t.Optional[t.Any] = None
    ss_factor: t.Optional[t.Any] = None


@attr.s(auto_attribs=True, frozen=True)
class PropsDataSeries:
    """Per-time-step data kept with `keep_iter_data`
    (qmc_exec/data/dmc.py:624-660)."""
    iter_props_blocks: t.Any
    ssf_blocks: t.Optional[np.ndarray] = None

    @property
    def props(self):
        p = self.iter_props_blocks
        return type(p)(*[np.hstack(getattr(p, f)) for f in p._fields])


@attr.s(auto_attribs=True, frozen=True)
class SamplingData:
    blocks: PropsDataBlocks
    series: t.Optional[PropsDataSeries] = None


# ---- HDF5 layout (qmc_exec/data/dmc.py:99-120, 192-211, 581-613, 683-735,
# 770-793): <group>/totals [, weight_totals]; ss_factor/{fdk_sqr_abs,fdk_real,
# fdk_imag}/...; blocks/{energy,weight,num_walkers[,density][,ss_factor]} ----
def _export_weighted(self, group):
    group.create_dataset('totals', data=self.totals)
    group.create_dataset('weight_totals', data=self.weight_totals)


def _import_weighted(cls, group):
    return cls(totals=group.get('totals')[()],
               weight_totals=group.get('weight_totals')[()])


def _export_unweighted(self, group):
    group.create_dataset('totals', data=self.totals)


def _import_unweighted(cls, group):
    return cls(totals=group.get('totals')[()])


PropBlocks.hdf5_export = _export_weighted
PropBlocks.from_hdf5_data = classmethod(_import_weighted)
UnWeightedPropBlocks.hdf5_export = _export_unweighted
UnWeightedPropBlocks.from_hdf5_data = classmethod(_import_unweighted)


def _ssf_export(self, group):
    self.fdk_sqr_abs_part.hdf5_export(group.require_group('fdk_sqr_abs'))
    self.fdk_real_part.hdf5_export(group.require_group('fdk_real'))
    self.fdk_imag_part.hdf5_export(group.require_group('fdk_imag'))


def _ssf_import(cls, group):
    return cls(SSFPartBlocks.from_hdf5_data(group.get('fdk_sqr_abs')),
               SSFPartBlocks.from_hdf5_data(group.get('fdk_real')),
               SSFPartBlocks.from_hdf5_data(group.get('fdk_imag')))


SSFBlocks.hdf5_export = _ssf_export
SSFBlocks.from_hdf5_data = classmethod(_ssf_import)


def _blocks_export(self, group):
    self.energy.hdf5_export(group.require_group('energy'))
    self.weight.hdf5_export(group.require_group('weight'))
    self.num_walkers.hdf5_export(group.require_group('num_walkers'))
    if self.density is not None:
        self.density.hdf5_export(group.require_group('density'))
    if self.ss_factor is not None:
        self.ss_factor.hdf5_export(group.require_group('ss_factor'))


def _blocks_import(cls, group):
    dens, ssf = group.get('density'), group.get('ss_factor')
    return cls(EnergyBlocks.from_hdf5_data(group.get('energy')),
               WeightBlocks.from_hdf5_data(group.get('weight')),
               NumWalkersBlocks.from_hdf5_data(group.get('num_walkers')),
               None if dens is None else DensityBlocks.from_hdf5_data(dens),
               None if ssf is None else SSFBlocks.from_hdf5_data(ssf))


PropsDataBlocks.hdf5_export = _blocks_export
PropsDataBlocks.from_hdf5_data = classmethod(_blocks_import)


def _sampling_export(self, group):
    # (the reference does not store the per-step series either)
    self.blocks.hdf5_export(group.require_group('blocks'))


def _sampling_import(cls, group):
    return cls(PropsDataBlocks.from_hdf5_data(group.get('blocks')))


SamplingData.hdf5_export = _sampling_export
SamplingData.from_hdf5_data = classmethod(_sampling_import)
