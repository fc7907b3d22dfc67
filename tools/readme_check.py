"""Runs the README quick-start snippet (smaller sizes) -- development check."""
import os
import sys
from itertools import islice
from math import pi

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phd_qmclib_amd import mrbp_qmc  # noqa

spec = mrbp_qmc.Spec(lattice_depth=5 * pi**2, lattice_ratio=1, interaction_strength=2,
                     boson_number=64, supercell_size=64, tbf_contact_cutoff=16)
vmc = mrbp_qmc.vmc.EnsembleSampling(spec, move_spread=0.125, num_chains=1 << 12, rng_seed=1)
vmc.init_random(seed=0)
for blk in islice(vmc.blocks(64), 2):
    print(blk.energy.mean() / 64, blk.accept_rate.mean())
dmc = mrbp_qmc.dmc.Sampling(spec, time_step=6.25e-4, max_num_walkers=4400,
                            target_num_walkers=4096, num_walkers_control_factor=0.5, rng_seed=1)
confs = np.zeros((4096, 2, 64)); confs[:, 0] = vmc.confs()
for blk in islice(dmc.blocks(dmc.build_state(confs), 32, 1), 2):
    print(blk.iter_props.energy.sum() / blk.iter_props.weight.sum() / 64)
print('README snippet OK')
