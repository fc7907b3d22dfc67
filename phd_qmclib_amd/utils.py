"""Small helpers shared by the samplings."""
import os
import time

import numpy as np

__all__ = ['get_random_rng_seed']


def get_random_rng_seed():
    """A fresh int31 seed for `rng_seed=None` (reference: utils.py:250-266:
    seeded from pid + wall-clock milliseconds so concurrent processes
    differ).  Uses a private RandomState: the global numpy stream that
    `Spec.init_get_sys_conf` draws from is left untouched."""
    i32_max = np.iinfo(np.int32).max
    rs = np.random.RandomState(int(os.getpid() + int(time.time() * 1000) % i32_max))
    return int(rs.randint(0, high=i32_max - 1, dtype=np.int64))
