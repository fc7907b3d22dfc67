"""Lieb-Liniger known answer vs time step / stale-energy quirk (dev tool)."""
import os
import sys
from math import pi

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phd_qmclib_amd.engine import DmcEnsemble, ModelEngine, VmcEnsemble  # noqa
from phd_qmclib_amd.mrbp_qmc import Spec  # noqa

n = 32
spec = Spec(lattice_depth=0.0, lattice_ratio=1, interaction_strength=4.0,
            boson_number=n, supercell_size=n, tbf_contact_cutoff=0.25 * n)
eng = ModelEngine(spec.cfc_spec)
W = 8192
v = VmcEnsemble(eng, W, 0.4, rng_seed=3)
v.set_state(n * np.random.RandomState(2).random_sample((W, n)))
v.run_block(400, sums=True)
out = v.run_block(200, sums=True)
print('VMC E/N', out['sum_energy'].sum() / (200 * W) / n, flush=True)
for dt in (2e-3, 1e-3, 5e-4, 2.5e-4):
    for fix in (False, True):
        d = DmcEnsemble(eng, dt, 9216, W, 0.5, rng_seed=4, fix_stale_energy=fix)
        d.set_state_from_vmc(v, W)
        neq = int(1.0 / dt)
        d.run_block(neq, read=False)
        ser = d.run_block(neq)
        print(f'dt={dt:g} fix_stale={fix}  E/N = '
              f'{ser.energy.sum() / ser.weight.sum() / n:.5f}  <nw>='
              f'{ser.num_walkers.mean():.0f}', flush=True)
        d.close()
