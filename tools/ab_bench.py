"""Quick A/B timing of the VMC and DMC step kernels (development tool).
usage: ab_bench.py [--bosons N] [--walkers W] [--steps K] [--equil E]
Prints ms per launch (HIP events around every launch) for the library /
environment it runs under (QMCWALK_LIB, QMCWALK_SHAPE, QMCWALK_OB_TABLE)."""
import argparse
import os
import sys
from math import pi

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phd_qmclib_amd.engine import DmcEnsemble, ModelEngine, VmcEnsemble  # noqa
from phd_qmclib_amd.mrbp_qmc import Spec  # noqa

ap = argparse.ArgumentParser()
ap.add_argument('--bosons', type=int, default=64)
ap.add_argument('--walkers', type=int, default=1 << 20)
ap.add_argument('--dmc-walkers', type=int, default=1 << 18)
ap.add_argument('--steps', type=int, default=48)
ap.add_argument('--equil', type=int, default=200)
ap.add_argument('--tag', default='')
ap.add_argument('--fast', action='store_true', help='float pair loop')
ap.add_argument('--no-dmc', action='store_true')
ap.add_argument('--start', default='random', choices=['random', 'lattice', 'stationary'],
                help='lattice: one particle per well, +-0.15 around its centre; '
                     'stationary: copies of a seed ensemble relaxed for 30 000 '
                     'steps (tools/_stationary.py), as bench.py')
a = ap.parse_args()
n = a.bosons
spec = Spec(lattice_depth=5 * pi ** 2, lattice_ratio=1, interaction_strength=2,
            boson_number=n, supercell_size=n, tbf_contact_cutoff=0.25 * n)
eng = ModelEngine(spec.cfc_spec, device=0, fast_math=a.fast)
rng = np.random.RandomState(1)
pos = n * rng.random_sample((a.walkers, n))
if a.start == 'stationary':
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _stationary import replicate, seed_configurations
    pos = replicate(seed_configurations(eng, spec, n), a.walkers)
elif a.start == 'lattice':
    pos = np.arange(n)[None, :] + 0.25 + 0.3 * (rng.random_sample((a.walkers, n)) - 0.5)
v = VmcEnsemble(eng, a.walkers, 0.25 * spec.well_width, rng_seed=1)
v.set_state(pos)
del pos
done = 0
while done < a.equil:
    v.run_block(50, sums=False)
    done += 50
eng.profile_begin(a.steps)
v.run_block(a.steps, sums=False)
nl, tot, mn, mx = eng.profile_end()
res = v.run_block(16)
e = res['sum_energy'].sum() / (16 * a.walkers * n)
acc = res['num_accepted'].sum() / (16 * a.walkers)
print(f'{a.tag:24s} VMC N={n} W={a.walkers}: {tot / nl:.4f} ms/launch '
      f'(min {mn:.4f} max {mx:.4f}) = {a.walkers / (tot / nl) / 1e3:.4g} steps/s'
      f'  E/N={e:.5f} acc={acc:.4f}', flush=True)
if a.no_dmc:
    sys.exit(0)
target = a.dmc_walkers
maxw = ((target * 512 // 480) + 255) // 256 * 256
d = DmcEnsemble(eng, 6.25e-4, maxw, target, 0.5, rng_seed=1)
d.set_state_from_vmc(v, target, replicate=True)
d.run_block(16, read=False)
eng.profile_begin(a.steps)
d.run_block(a.steps, read=False)
nl, tot, mn, mx = eng.profile_end()
ser = d.read_series(a.steps)
print(f'{a.tag:24s} DMC N={n} W={target}: {tot / nl:.4f} ms/evolve '
      f'(min {mn:.4f} max {mx:.4f}) = '
      f'{ser.num_walkers.mean() / (tot / nl) / 1e3:.4g} steps/s  '
      f'E/N={ser.energy.sum() / ser.weight.sum() / n:.5f}', flush=True)
