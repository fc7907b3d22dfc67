"""On-the-fly power-of-two reblocking of serially correlated data.

Mirrors the quantities of the reference's `stats.reblock` (reblock.py:113-226,
436-604, 651-756): block means are formed hierarchically (a block of size 2B is
the mean of two consecutive blocks of size B), and for every order the table
keeps (block_size, sum of block means, sum of squared block means, number of
blocks).  From it: variances of the block means, integrated autocorrelation
times tau(B) = B var_B / (2 var_1), the optimal block size (smallest B with
B^3 > 8 n tau(B)^2), the effective sample size n / (2 tau_opt) and the error
of the mean.  Host-side numpy: the inputs are a few hundred block totals.
"""
import typing as t
from math import floor, log, sqrt
from warnings import warn

import attr
import numpy as np

__all__ = ['OTFObject', 'OTFSet', 'otf_data_dtype', 'on_the_fly_obj_create',
           'on_the_fly_obj_data_order']

otf_data_dtype = np.dtype([
    ('BLOCK_SIZE', np.int64),
    ('MEANS', np.float64),
    ('MEANS_SQR', np.float64),
    ('NUM_BLOCKS', np.int64)
])


def on_the_fly_obj_data_order(source_data) -> int:
    """Largest order of the table (reblock.py:447-457; the float formula is
    the reference's own)."""
    return int(floor(log(len(source_data)) / log(2)))


def on_the_fly_obj_create(source_data) -> np.ndarray:
    """The reblocking table of a 1-D series, or one table per column of a 2-D
    array [samples, columns] (reblock.py:479-604)."""
    x = np.asarray(source_data, dtype=np.float64)
    if x.ndim not in (1, 2):
        raise ValueError('source_data must be a 1d or 2d array')
    one_d = x.ndim == 1
    if one_d:
        x = x[:, np.newaxis]
    max_order = on_the_fly_obj_data_order(x)
    ncols = x.shape[1]
    table = np.zeros((ncols, max_order + 1), dtype=otf_data_dtype)
    level = x
    for order in range(max_order + 1):
        if order:
            m = len(level) // 2
            level = (level[0:2 * m:2] + level[1:2 * m:2]) / 2
        table['BLOCK_SIZE'][:, order] = 1 << order
        table['NUM_BLOCKS'][:, order] = len(level)
        # sequential accumulation, like the reference's running sums
        table['MEANS'][:, order] = np.add.accumulate(level, axis=0)[-1]
        table['MEANS_SQR'][:, order] = np.add.accumulate(level * level,
                                                         axis=0)[-1]
    return table[0] if one_d else table


@attr.s(auto_attribs=True, frozen=True)
class OTFObject:
    """Reblocking analysis over an on-the-fly table (reblock.py:651-756)."""

    source_data: np.ndarray
    min_num_blocks: t.Optional[int] = 2
    var_ddof: int = 1

    def __attrs_post_init__(self):
        data = self.source_data
        if not data.dtype == otf_data_dtype:
            raise TypeError("source_data is not a reblocking table.")
        if data.ndim != 1:
            raise ValueError("source_data must be a 1d array")
        object.__setattr__(self, 'var_ddof', 1)
        mnb = self.min_num_blocks or 2
        if mnb < 2:
            raise ValueError('the minimum number of blocks of the reblocking '
                             'is two')
        object.__setattr__(self, 'min_num_blocks', mnb)
        keep = data['NUM_BLOCKS'] >= mnb
        if not np.count_nonzero(keep):
            raise ValueError('the source data is empty for the requested '
                             'minimum number of blocks.')
        object.__setattr__(self, 'source_data', data[keep])

    @classmethod
    def from_non_obj_data(cls, seq, min_num_blocks: int = None):
        return cls(on_the_fly_obj_create(seq), min_num_blocks=min_num_blocks)

    @property
    def block_sizes(self):
        return self.source_data['BLOCK_SIZE']

    @property
    def num_blocks(self):
        return self.source_data['NUM_BLOCKS']

    @property
    def size(self):
        return self.num_blocks[0]

    @property
    def means(self):
        return self.source_data['MEANS'] / self.num_blocks

    @property
    def vars(self):
        nb = self.num_blocks
        means_sqr = self.source_data['MEANS_SQR'] / nb
        return nb * (means_sqr - self.means ** 2) / (nb - self.var_ddof)

    @property
    def mean(self):
        return self.means[0]

    @property
    def var(self):
        return self.vars[0]

    @property
    def errors(self):
        return np.sqrt(self.vars / self.num_blocks)

    @property
    def iac_times(self):
        return 0.5 * self.block_sizes * self.vars / self.var

    @property
    def opt_block_size(self):
        """Smallest B with B^3 > 2 n (2 tau)^2 (reblock.py:175-191)."""
        bs = self.block_sizes
        ok = bs ** 3 > 8 * self.size * self.iac_times ** 2
        if not np.count_nonzero(ok):
            warn("the optimum block size criterion is not satisfied by "
                 "any of the autocorrelation times. The maximum block "
                 "size will be treated as the optimal one. You may try "
                 "to gather more data to suppress this warning.",
                 RuntimeWarning)
            return bs.max()
        return bs[ok].min()

    @property
    def opt_iac_time(self):
        return self.iac_times[self.block_sizes == self.opt_block_size][0]

    @property
    def eff_size(self):
        return self.size / (2 * self.opt_iac_time)

    @property
    def mean_eff_error(self):
        return sqrt(self.var / self.eff_size)


@attr.s(auto_attribs=True, frozen=True)
class OTFSet:
    """Reblocking of several series at once: one table per column
    (reblock.py:229-323, 759-923).  Every property has a leading column
    axis."""

    source_data: np.ndarray
    min_num_blocks: t.Optional[int] = 2
    var_ddof: int = 1

    def __attrs_post_init__(self):
        data = self.source_data
        if not data.dtype == otf_data_dtype:
            raise TypeError("source_data is not a reblocking table.")
        if data.ndim != 2:
            raise ValueError("source_data must be a 2d array")
        object.__setattr__(self, 'var_ddof', 1)
        mnb = self.min_num_blocks or 2
        if mnb < 2:
            raise ValueError('the minimum number of blocks of the reblocking '
                             'is two')
        object.__setattr__(self, 'min_num_blocks', mnb)
        keep = data['NUM_BLOCKS'][0, :] >= mnb
        if not np.count_nonzero(keep):
            raise ValueError('the source data is empty for the requested '
                             'minimum number of blocks.')
        object.__setattr__(self, 'source_data', data[:, keep])

    @classmethod
    def from_non_obj_data(cls, seq, min_num_blocks: int = None):
        seq = np.asarray(seq)
        if seq.ndim == 1:
            seq = seq[:, np.newaxis]
        return cls(on_the_fly_obj_create(seq), min_num_blocks=min_num_blocks)

    @property
    def block_sizes(self):
        return self.source_data['BLOCK_SIZE']

    @property
    def num_blocks(self):
        return self.source_data['NUM_BLOCKS']

    @property
    def size(self):
        return self.num_blocks[:, 0]

    @property
    def means(self):
        return self.source_data['MEANS'] / self.num_blocks

    @property
    def vars(self):
        nb = self.num_blocks
        means_sqr = self.source_data['MEANS_SQR'] / nb
        return nb * (means_sqr - self.means ** 2) / (nb - self.var_ddof)

    @property
    def mean(self):
        return self.means[:, 0]

    @property
    def var(self):
        return self.vars[:, 0]

    @property
    def errors(self):
        return np.sqrt(self.vars / self.num_blocks)

    @property
    def iac_times(self):
        return 0.5 * self.block_sizes * self.vars / self.var[:, np.newaxis]

    @property
    def opt_block_size(self):
        bs = self.block_sizes
        ok = bs ** 3 > 8 * self.size[:, np.newaxis] * self.iac_times ** 2
        out = []
        for row, sel in enumerate(ok):
            valid = bs[row, sel]
            if not np.count_nonzero(valid):
                warn("the optimum block size criterion is not satisfied by "
                     "any of the autocorrelation times. The maximum block "
                     "size will be treated as the optimal one. You may try "
                     "to gather more data to suppress this warning.",
                     RuntimeWarning)
                out.append(bs.max())
            else:
                out.append(valid.min())
        return np.array(out)

    @property
    def opt_iac_time(self):
        sel = self.block_sizes == self.opt_block_size[:, np.newaxis]
        return np.array([self.iac_times[r, m][0] for r, m in enumerate(sel)])

    @property
    def eff_size(self):
        return self.size / (2 * self.opt_iac_time)

    @property
    def mean_eff_error(self):
        return np.sqrt(self.var / self.eff_size)

    def __len__(self):
        return self.source_data.shape[0]

    def __getitem__(self, index) -> OTFObject:
        return OTFObject(self.source_data[index],
                         min_num_blocks=self.min_num_blocks)
