"""Configurations of the benchmark box in the STATIONARY state of the VMC chain
(development tools; bench.py has its own copy of the recipe and explains it).

From a uniform random start the chain needs ~40 000 Metropolis steps until
E/N (15.395 at N = 64) and the acceptance ratio (0.450) stop moving; from one
particle per lattice well ~20 000.  A seed ensemble of a few thousand chains
does that in a second; the full ensemble starts from copies of its
configurations and a few hundred steps take the copies apart."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phd_qmclib_amd.engine import VmcEnsemble  # noqa: E402


def seed_configurations(eng, spec, n, seeds=4096, steps=30000, spread=None,
                        rng_seed=7):
    """-> [seeds, n] positions after `steps` steps from one particle per well."""
    rng = np.random.RandomState(rng_seed)
    ww = spec.well_width
    pos = (np.arange(n)[None, :] + 0.5 * ww +
           0.6 * ww * (rng.random_sample((seeds, n)) - 0.5))
    v = VmcEnsemble(eng, seeds, 0.25 * ww if spread is None else spread,
                    rng_seed=2)
    v.set_state(pos)
    done = 0
    while done < steps:
        b = min(128, steps - done)
        v.run_block(b, sums=False)
        done += b
    pos = v.get_state()[0]
    v.close()
    return pos


def replicate(seed_pos, chains):
    """[chains, n]: the seed configurations reused cyclically."""
    reps = -(-chains // len(seed_pos))
    return np.ascontiguousarray(np.tile(seed_pos, (reps, 1))[:chains])
