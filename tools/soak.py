"""Soak run of the production kernels at the benchmarked sizes (development
tool; record in profiles/): thousands of steps of the VMC step at N = 64, 2^20
chains, and of the DMC step at N = 64 (2^18 walkers) and N = 128 (2^20), with
the invariants a production run relies on checked along the way -- finite
sums, energies and acceptance inside their windows, the population inside its
cap and near its target, unit weights after branching, E_t = sum of the yielded
energies -- and the device's count of walkers that left the sorted-row pair sums.
usage: soak.py [--vmc-steps K] [--dmc-steps K]"""
import argparse
import os
import sys
import time
from math import pi

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phd_qmclib_amd import _lib  # noqa: E402
from phd_qmclib_amd.engine import DmcEnsemble, ModelEngine, VmcEnsemble  # noqa
from phd_qmclib_amd.mrbp_qmc import Spec  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--vmc-steps', type=int, default=4096)
ap.add_argument('--dmc-steps', type=int, default=2048)
a = ap.parse_args()


def box(n):
    return Spec(lattice_depth=5 * pi ** 2, lattice_ratio=1, interaction_strength=2,
                boson_number=n, supercell_size=n, tbf_contact_cutoff=0.25 * n)


print('library', _lib.LIB_PATH, 'kernels', _lib.source_hash(), flush=True)
for n, W, Wd in ((64, 1 << 20, 1 << 18), (128, 1 << 18, 1 << 20)):
    spec = box(n)
    eng = ModelEngine(spec.cfc_spec, device=0)
    rng = np.random.RandomState(n)
    v = VmcEnsemble(eng, W, 0.25 * spec.well_width, rng_seed=7)
    pos = np.empty((W, n))
    for lo in range(0, W, 1 << 16):
        pos[lo:lo + (1 << 16)] = n * rng.random_sample((min(1 << 16, W - lo), n))
    v.set_state(pos)
    del pos
    eng.general_path_walkers(reset=True)
    t0 = time.time()
    done, blk = 0, 256
    print(f'== VMC N={n}, {W} chains, {a.vmc_steps} steps from a uniform random start', flush=True)
    while done < a.vmc_steps:
        out = v.run_block(blk)
        done += blk
        e = out['sum_energy'] / blk / n
        acc = out['num_accepted'].sum() / (blk * W)
        gp = eng.general_path_walkers()
        assert np.all(np.isfinite(out['sum_energy'])) and np.all(np.isfinite(out['sum_energy2']))
        assert 15.0 < e.mean() < 17.5 and 0.25 < acc < 0.6, (e.mean(), acc)
        print(f'  steps {done:6d}  E/N {e.mean():.5f} +- {e.std() / W ** 0.5:.5f}  acceptance {acc:.4f}  '
              f'general-path walkers in this block {gp} of {blk * W} ({gp / (blk * W):.2e})', flush=True)
    print(f'  {time.time() - t0:.1f} s', flush=True)
    target = Wd
    maxw = ((target * 512 // 480) + 255) // 256 * 256
    d = DmcEnsemble(eng, 6.25e-4, maxw, target, 0.5, rng_seed=11)
    d.set_state_from_vmc(v, target, replicate=True)
    v.close()
    eng.general_path_walkers(reset=True)
    print(f'== DMC N={n}, target {target} / cap {maxw}, {a.dmc_steps} steps from the VMC chains', flush=True)
    done, blk = 0, 128
    t0 = time.time()
    while done < a.dmc_steps:
        ser = d.run_block(blk)
        done += blk
        gp = eng.general_path_walkers()
        nw = ser.num_walkers.astype(np.float64)
        assert np.all(np.isfinite(ser.energy)) and np.all(np.isfinite(ser.ref_energy))
        assert np.array_equal(ser.weight, nw)                 # unit weights after branching
        assert nw.max() <= maxw and nw.min() > 0.9 * target
        e = ser.energy.sum() / ser.weight.sum() / n
        assert 15.0 < e < 16.2, e
        print(f'  steps {done:6d}  E/N {e:.5f}  walkers {int(nw.min())}..{int(nw.max())}  E_ref/N {ser.ref_energy[-1] / n:.4f}  '
              f'general-path walkers {gp} of {int(nw.sum())} ({gp / nw.sum():.2e})', flush=True)
    st = d.get_scalars()
    print(f'  {time.time() - t0:.1f} s; accumulated E/N {st[3] / n:.5f}', flush=True)
    d.close()
    eng.close()
print('soak ok')
