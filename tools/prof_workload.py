"""Small fixed workloads for rocprofv3 counter passes (development tool).
usage: prof_workload.py {vmc|dmc} [--bosons N] [--walkers W] [--steps K]"""
import argparse
import os
import sys
from math import pi

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phd_qmclib_amd.engine import DmcEnsemble, ModelEngine, VmcEnsemble  # noqa
from phd_qmclib_amd.mrbp_qmc import Spec  # noqa

ap = argparse.ArgumentParser()
ap.add_argument('kind', choices=['vmc', 'dmc'])
ap.add_argument('--bosons', type=int, default=64)
ap.add_argument('--walkers', type=int, default=1 << 18)
ap.add_argument('--steps', type=int, default=8)
ap.add_argument('--launches', type=int, default=2)
ap.add_argument('--equil', type=int, default=0, help='untimed steps first (VMC)')
ap.add_argument('--fast', action='store_true', help='float pair loop')
ap.add_argument('--start-file', default='',
                help='.npy of seed configurations (tools/make_stationary.py): the '
                     'walkers start from copies of them instead of a uniform '
                     'random row, and --equil steps take the copies apart')
a = ap.parse_args()
n = a.bosons
spec = Spec(lattice_depth=5 * pi ** 2, lattice_ratio=1, interaction_strength=2,
            boson_number=n, supercell_size=n, tbf_contact_cutoff=0.25 * n)
eng = ModelEngine(spec.cfc_spec, device=0, fast_math=a.fast)
rng = np.random.RandomState(1)
pos = n * rng.random_sample((a.walkers, n))
if a.start_file:
    seed = np.load(a.start_file)
    pos = np.ascontiguousarray(
        np.tile(seed, (-(-a.walkers // len(seed)), 1))[:a.walkers])
if a.kind == 'vmc':
    v = VmcEnsemble(eng, a.walkers, 0.25 * spec.well_width, rng_seed=1)
    v.set_state(pos)
    done = 0
    while done < a.equil:
        v.run_block(min(50, a.equil - done), sums=False)
        done += 50
    for _ in range(a.launches):
        v.run_block(a.steps, sums=False)
    eng.sync()
else:
    maxw = ((a.walkers * 512 // 480) + 255) // 256 * 256
    d = DmcEnsemble(eng, 6.25e-4, maxw, a.walkers, 0.5, rng_seed=1)
    if a.equil:
        # walkers from equilibrated VMC chains, as in bench.py
        v = VmcEnsemble(eng, a.walkers, 0.25 * spec.well_width, rng_seed=1)
        v.set_state(pos)
        done = 0
        while done < a.equil:
            v.run_block(min(50, a.equil - done), sums=False)
            done += 50
        d.set_state_from_vmc(v, a.walkers)
        v.close()
    else:
        d.set_state(pos)
    d.run_block(a.steps, read=False)
    eng.sync()
from phd_qmclib_amd import _lib  # noqa: E402
print('kernel_source_sha=' + _lib.source_hash())
print('done')
