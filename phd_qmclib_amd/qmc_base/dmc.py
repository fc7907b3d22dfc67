"""Data contracts of the DMC sampling (reference: qmc_base/dmc.py:98-168)."""
import enum
import typing as t

import numpy as np

__all__ = ['BranchingSpec', 'StateProps', 'StateData', 'State', 'PropsData',
           'SamplingBlock', 'SamplingStateDataBlock', 'iter_props_dtype',
           'IterProp', 'StateProp', 'SSFPartSlot']


@enum.unique
class StateProp(str, enum.Enum):
    ENERGY = 'ENERGY'
    WEIGHT = 'WEIGHT'
    MASK = 'MASK'


@enum.unique
class IterProp(str, enum.Enum):
    ENERGY = 'ENERGY'
    WEIGHT = 'WEIGHT'
    NUM_WALKERS = 'NUM_WALKERS'
    REF_ENERGY = 'REF_ENERGY'
    ACCUM_ENERGY = 'ACCUM_ENERGY'


@enum.unique
class SSFPartSlot(enum.IntEnum):
    FDK_SQR_ABS = 0
    FDK_REAL = 1
    FDK_IMAG = 2


class BranchingSpec(t.NamedTuple):
    """The cloning table (qmc_base/dmc.py:98-101)."""
    cloning_factor: np.ndarray
    cloning_ref: np.ndarray


class StateProps(t.NamedTuple):
    """Per-walker properties (qmc_base/dmc.py:104-108)."""
    energy: np.ndarray
    weight: np.ndarray
    mask: np.ndarray


class StateData(t.NamedTuple):
    confs: np.ndarray
    props: StateProps


class State(t.NamedTuple):
    """A DMC state (qmc_base/dmc.py:117-127)."""
    confs: np.ndarray
    props: StateProps
    energy: float
    weight: float
    num_walkers: int
    ref_energy: float
    accum_energy: float
    max_num_walkers: int
    branching_spec: t.Optional[BranchingSpec] = None


iter_props_dtype = np.dtype([
    (IterProp.ENERGY.value, np.float64),
    (IterProp.WEIGHT.value, np.float64),
    (IterProp.NUM_WALKERS.value, np.uint64),
    (IterProp.REF_ENERGY.value, np.float64),
    (IterProp.ACCUM_ENERGY.value, np.float64)
])


class PropsData(t.NamedTuple):
    """Per-time-step series of a block (qmc_base/dmc.py:130-143)."""
    energy: np.ndarray
    weight: np.ndarray
    num_walkers: np.ndarray
    ref_energy: np.ndarray
    accum_energy: np.ndarray

    def as_record(self):
        fields = (self.energy, self.weight, self.num_walkers,
                  self.ref_energy, self.accum_energy)
        return np.array(list(zip(*fields)), dtype=iter_props_dtype)


class SamplingBlock(t.NamedTuple):
    """One block of time steps (qmc_base/dmc.py:146-152)."""
    iter_props: PropsData
    iter_density: t.Optional[np.ndarray]
    iter_ssf: t.Optional[np.ndarray] = None
    last_state: t.Optional[State] = None


class SamplingStateDataBlock(t.NamedTuple):
    confs: np.ndarray
    props: StateProps
    iter_props: PropsData
