"""Per-step latency at the reference's typical sizes (development tool)."""
import os
import sys
import time
from math import pi

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phd_qmclib_amd.engine import DmcEnsemble, ModelEngine, VmcEnsemble  # noqa
from phd_qmclib_amd.mrbp_qmc import Spec  # noqa

for n, W in ((16, 480), (16, 4096), (64, 480), (64, 4096), (64, 16384)):
    spec = Spec(lattice_depth=5 * pi ** 2, lattice_ratio=1,
                interaction_strength=2, boson_number=n, supercell_size=n,
                tbf_contact_cutoff=0.25 * n)
    eng = ModelEngine(spec.cfc_spec, device=0)
    pos = n * np.random.RandomState(1).random_sample((W, n))
    maxw = ((W * 512 // 480) + 255) // 256 * 256
    d = DmcEnsemble(eng, 1e-3, maxw, W, 0.5, rng_seed=1)
    d.set_state(pos)
    d.run_block(512)          # (a captured graph is built on first use)
    eng.sync()
    t0 = time.perf_counter()
    d.run_block(512)
    eng.sync()
    dt = time.perf_counter() - t0
    v = VmcEnsemble(eng, 1, 0.125, rng_seed=1)
    v.set_state(pos[:1])
    v.run_block(64, series=True)
    t0 = time.perf_counter()
    v.run_block(2048, series=True)
    vt = time.perf_counter() - t0
    print(f'N={n:3d} W={W:6d}  DMC {dt / 512 * 1e6:7.1f} us/step   '
          f'VMC single chain {vt / 2048 * 1e6:6.1f} us/step', flush=True)
    d.close(); v.close(); eng.close()
