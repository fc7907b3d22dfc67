"""Cost of the DMC estimators next to the plain time step (development tool).
usage: est_bench.py [--bosons N] [--walkers W] [--steps K] [--modes M] [--bins B]"""
import argparse
import os
import sys
import time
from math import pi

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phd_qmclib_amd.engine import DmcEnsemble, ModelEngine  # noqa
from phd_qmclib_amd.mrbp_qmc import Spec  # noqa

ap = argparse.ArgumentParser()
ap.add_argument('--bosons', type=int, default=64)
ap.add_argument('--walkers', type=int, default=1 << 17)
ap.add_argument('--steps', type=int, default=16)
ap.add_argument('--modes', type=int, default=64)
ap.add_argument('--bins', type=int, default=128)
a = ap.parse_args()
n = a.bosons
spec = Spec(lattice_depth=5 * pi ** 2, lattice_ratio=1, interaction_strength=2,
            boson_number=n, supercell_size=n, tbf_contact_cutoff=0.25 * n)
eng = ModelEngine(spec.cfc_spec, device=0)
pos = n * np.random.RandomState(1).random_sample((a.walkers, n))
maxw = ((a.walkers * 512 // 480) + 255) // 256 * 256


def timed(tag, **est):
    d = DmcEnsemble(eng, 6.25e-4, maxw, a.walkers, 0.5, rng_seed=1)
    d.set_state(pos)
    if est:
        d.set_estimators(**est)
        run = lambda k: d.run_block_est(k, True)
    else:
        run = lambda k: d.run_block(k)
    run(4)
    eng.sync()
    t0 = time.perf_counter()
    run(a.steps)
    eng.sync()
    dt = time.perf_counter() - t0
    print(f'{tag:28s} {dt / a.steps * 1e3:8.3f} ms/step', flush=True)
    d.close()
    return dt


base = timed('plain')
timed(f'ssf mixed M={a.modes}', num_modes=a.modes)
timed(f'ssf pure  M={a.modes}', num_modes=a.modes, ssf_pure=True, ssf_pfw=8)
timed(f'density mixed B={a.bins}', num_bins=a.bins)
timed(f'density pure  B={a.bins}', num_bins=a.bins, dens_pure=True, dens_pfw=8)
