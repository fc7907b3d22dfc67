"""Data contracts of the VMC sampling (reference: qmc_base/vmc.py:128-161).

The same named tuples the reference's generators yield, so a caller written
against `phd_qmclib.qmc_base.vmc` reads the same fields.
"""
import enum
import typing as t

import numpy as np

__all__ = ['State', 'PropsData', 'SamplingBlock', 'SamplingStateDataBlock',
           'EnsembleBlock', 'STAT_ACCEPTED', 'STAT_REJECTED', 'SSFPartSlot',
           'IterProp', 'StateProp', 'RandDisplaceStat']


class RandDisplaceStat(enum.IntEnum):
    REJECTED = 0
    ACCEPTED = 1


STAT_REJECTED = int(RandDisplaceStat.REJECTED)
STAT_ACCEPTED = int(RandDisplaceStat.ACCEPTED)


@enum.unique
class SSFPartSlot(enum.IntEnum):
    """Contributions to the static structure factor (qmc_base/vmc.py:60-72)."""
    FDK_SQR_ABS = 0
    FDK_REAL = 1
    FDK_IMAG = 2


@enum.unique
class StateProp(str, enum.Enum):
    WF_ABS_LOG = 'WF_ABS_LOG'
    MOVE_STAT = 'MOVE_STAT'


@enum.unique
class IterProp(str, enum.Enum):
    WF_ABS_LOG = 'WF_ABS_LOG'
    ENERGY = 'ENERGY'
    MOVE_STAT = 'MOVE_STAT'


class State(t.NamedTuple):
    """What the VMC generator yields every step (qmc_base/vmc.py:128-132)."""
    sys_conf: np.ndarray
    wf_abs_log: float
    move_stat: int


class PropsData(t.NamedTuple):
    """Per-step series of a block (qmc_base/vmc.py:135-139)."""
    wf_abs_log: np.ndarray
    energy: np.ndarray
    move_stat: np.ndarray


class SamplingBlock(t.NamedTuple):
    """One block of the Markov chain (qmc_base/vmc.py:142-147)."""
    iter_props: PropsData
    iter_ssf: t.Optional[np.ndarray]
    accept_rate: float
    last_state: t.Optional[State] = None


class SamplingStateDataBlock(t.NamedTuple):
    """A block with every configuration kept (qmc_base/vmc.py:150-155)."""
    confs: np.ndarray
    props: PropsData
    accept_rate: float
    last_state: t.Optional[State] = None


class EnsembleBlock(t.NamedTuple):
    """Extension (no reference counterpart): one block of W independent
    chains advanced together on the GPU; per-chain block sums only."""
    sum_energy: np.ndarray        # [W]  sum over the block's steps
    sum_energy2: np.ndarray       # [W]
    num_accepted: np.ndarray      # [W]
    num_steps: int

    @property
    def energy(self):
        """Block-mean local energy of every chain."""
        return self.sum_energy / self.num_steps

    @property
    def accept_rate(self):
        return self.num_accepted / self.num_steps
