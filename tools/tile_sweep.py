"""N = 512 stress (BASELINE.json configs[4]): throughput of the batch
evaluation (log|psi|, energy, drift), the VMC step and the DMC step for the
library / precision it runs under (development tool; tools/tile_sweep.sh).
usage: tile_sweep.py [--fast] [--tag T] [--walkers W]"""
import argparse
import os
import sys
from math import pi

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phd_qmclib_amd.engine import (DeviceBuffer, DmcEnsemble, ModelEngine,  # noqa
                                   VmcEnsemble)
from phd_qmclib_amd.mrbp_qmc import Spec  # noqa

ap = argparse.ArgumentParser()
ap.add_argument('--bosons', type=int, default=512)
ap.add_argument('--walkers', type=int, default=1 << 14)
ap.add_argument('--fast', action='store_true')
ap.add_argument('--tag', default='')
a = ap.parse_args()
n, W = a.bosons, a.walkers
spec = Spec(lattice_depth=5 * pi ** 2, lattice_ratio=1, interaction_strength=2,
            boson_number=n, supercell_size=n, tbf_contact_cutoff=0.25 * n)
eng = ModelEngine(spec.cfc_spec, device=0, fast_math=a.fast)
pairs = n * (n - 1) / 2
rng = np.random.RandomState(1)
pos = n * rng.random_sample((W, n))
# K1 alone: the batch evaluation on resident buffers
dpos = DeviceBuffer((W, n)).upload(pos)
dwf, den = DeviceBuffer((W,)), DeviceBuffer((W,))
ddr = DeviceBuffer((W, n))
for _ in range(2):
    eng.evaluate_dev(W, dpos.ptr, dwf.ptr, den.ptr, 0, ddr.ptr)
eng.sync()
eng.timer_start()
for _ in range(4):
    eng.evaluate_dev(W, dpos.ptr, dwf.ptr, den.ptr, 0, ddr.ptr)
ev_ms = eng.timer_stop() / 4
v = VmcEnsemble(eng, W, 0.125, rng_seed=1)
v.set_state(pos)
v.run_block(40, sums=False)
eng.sync()
eng.profile_begin(8)
v.run_block(8, sums=False)
nl, tot, mn, mx = eng.profile_end()
vm = tot / nl
maxw = ((W * 512 // 480) + 255) // 256 * 256
d = DmcEnsemble(eng, 6.25e-4, maxw, W, 0.5, rng_seed=1)
d.set_state_from_vmc(v, W)
d.run_block(4, read=False)
eng.profile_begin(8)
d.run_block(8, read=False)
nl, tot, mn, mx = eng.profile_end()
ser = d.read_series(8)
dm = tot / nl
nw = float(ser.num_walkers.mean())
print(f'{a.tag:16s} {"f32" if eng.fast_math else "f64"} N={n} W={W}: '
      f'evaluate {ev_ms:7.3f} ms ({W * pairs / ev_ms / 1e6:6.1f} Gpair/s)  '
      f'vmc_step {vm:7.3f} ms ({W * pairs / vm / 1e6:6.1f})  '
      f'dmc_evolve {dm:7.3f} ms ({nw * pairs / dm / 1e6:6.1f})  '
      f'E/N={ser.energy.sum() / ser.weight.sum() / n:.4f}', flush=True)
