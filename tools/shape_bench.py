"""Throughput of the DMC step and VMC step for several N (development tool).
usage: shape_bench.py [--equil E] [--random] [N ...]
The ensembles are in the stationary state of the VMC chain (tools/_stationary.py:
a seed ensemble of 2048 chains, 20 000 steps from one particle per well,
copies + --equil steps), as in bench.py; --random: --equil steps after a
uniform random start, as rounds 2-3 measured."""
import os, sys, time
from math import pi
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phd_qmclib_amd.engine import DmcEnsemble, ModelEngine, VmcEnsemble
from phd_qmclib_amd.mrbp_qmc import Spec

args = sys.argv[1:]
equil = 200
if '--equil' in args:
    i = args.index('--equil')
    equil = int(args[i + 1])
    del args[i:i + 2]
random_start = '--random' in args
if random_start:
    args.remove('--random')
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _stationary import replicate, seed_configurations  # noqa: E402
sizes = [int(a) for a in args] or [16, 24, 37, 48, 63, 64, 100, 128, 256, 512]
print('# ' + ('uniform random start' if random_start else
              'stationary ensembles (2048 seed chains x 20000 steps)') +
      f' + {equil} steps', flush=True)
for n in sizes:
    W = max(1 << 13, min(1 << 20, (1 << 24) // n // 64 * 64))
    spec = Spec(lattice_depth=5 * pi ** 2, lattice_ratio=1, interaction_strength=2,
                boson_number=n, supercell_size=n, tbf_contact_cutoff=0.25 * n)
    eng = ModelEngine(spec.cfc_spec, device=0)
    rng = np.random.RandomState(1)
    if random_start:
        pos = n * rng.random_sample((W, n))
    else:
        pos = replicate(seed_configurations(eng, spec, n, seeds=2048,
                                            steps=20000, spread=0.125), W)
    v = VmcEnsemble(eng, W, 0.125, rng_seed=1)
    v.set_state(pos)
    v.run_block(equil, sums=False); eng.sync()
    eng.timer_start(); v.run_block(8, sums=False); ms = eng.timer_stop()
    vr = W * 8 / (ms * 1e-3)
    maxw = ((W * 512 // 480) + 255) // 256 * 256
    d = DmcEnsemble(eng, 6.25e-4, maxw, W, 0.5, rng_seed=1)
    d.set_state(v.get_state()[0])
    d.run_block(4, read=False); eng.sync()
    eng.timer_start(); d.run_block(8, read=False); ms = eng.timer_stop()
    ser = d.read_series(8)
    dr = float(ser.num_walkers.sum()) / (ms * 1e-3)
    pairs = n * (n - 1) / 2
    res = v.run_block(8)
    acc = res['num_accepted'].sum() / (8 * W)
    print(f'N={n:4d} W={W:8d}  VMC {vr:10.3e} steps/s ({vr*pairs:9.3e} pairs/s) acc {acc:.3f}   DMC {dr:10.3e} steps/s ({dr*pairs:9.3e} pairs/s)', flush=True)
    d.close(); v.close(); eng.close()
