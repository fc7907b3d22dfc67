"""Dynamic per-section attribution of the walker kernels (development tool;
VERDICT r2 item 4).

Needs a diagnostic library built with -DQMC_TIMING:
    tools/build_variant.sh timing "-DQMC_TIMING"
    QMCWALK_LIB=build/variants/timing/libqmcwalk.so python tools/section_times.py

Every section mark of the kernels (QMC_SECTION in csrc/) reads the shader clock;
the time since the wavefront's previous mark is booked to the section that ends
there.  Printed: the share of the wavefronts' lifetime per section and the
cycles per visit.  With 8 wavefronts per SIMD a wavefront's lifetime is about
8x its share of the SIMD, so shares -- not absolute cycles -- are the result.
The stamps themselves cost ~100 cycles each (an s_memtime round trip and two
atomics from lane 0); the diagnostic kernel runs ~5 % slower than the shipped
one."""
import argparse
import os
import sys
from math import pi

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phd_qmclib_amd.engine import DmcEnsemble, ModelEngine, VmcEnsemble  # noqa
from phd_qmclib_amd.mrbp_qmc import Spec  # noqa


def table(title, prof, units):
    tot = sum(c for c, _ in prof.values())
    print(f'== {title}: {tot / units:.0f} cycles of wavefront lifetime per '
          f'{title.split()[0]} step ({units} steps)')
    print(f'  {"section":32s} {"share":>7s} {"cycles/visit":>13s} '
          f'{"visits/step":>12s} {"cycles/step":>12s}')
    for name, (c, v) in prof.items():
        print(f'  {name:32s} {100 * c / tot:6.1f}% {c / v:13.0f} '
              f'{v / units:12.3f} {c / units:12.0f}')


ap = argparse.ArgumentParser()
ap.add_argument('--bosons', type=int, default=64)
ap.add_argument('--walkers', type=int, default=1 << 18)
ap.add_argument('--steps', type=int, default=32)
ap.add_argument('--equil', type=int, default=300)
ap.add_argument('--start-file', default='',
                help='.npy of stationary seed configurations')
a = ap.parse_args()
n = a.bosons
spec = Spec(lattice_depth=5 * pi ** 2, lattice_ratio=1, interaction_strength=2,
            boson_number=n, supercell_size=n, tbf_contact_cutoff=0.25 * n)
eng = ModelEngine(spec.cfc_spec, device=0)
rng = np.random.RandomState(1)
pos = n * rng.random_sample((a.walkers, n))
if a.start_file:                     # (tools/make_stationary.py)
    seed = np.load(a.start_file)
    pos = np.ascontiguousarray(
        np.tile(seed, (-(-a.walkers // len(seed)), 1))[:a.walkers])
v = VmcEnsemble(eng, a.walkers, 0.25 * spec.well_width, rng_seed=1)
v.set_state(pos)
done = 0
while done < a.equil:
    v.run_block(50, sums=False)
    done += 50
eng.section_profile(reset=True)
v.run_block(a.steps, sums=False)
prof = eng.section_profile(reset=True)
res = v.run_block(16)
acc = res['num_accepted'].sum() / (16 * a.walkers)
table(f'VMC N={n} (acceptance {acc:.3f})', prof, a.walkers * a.steps)

maxw = ((a.walkers * 512 // 480) + 255) // 256 * 256
d = DmcEnsemble(eng, 6.25e-4, maxw, a.walkers, 0.5, rng_seed=1)
d.set_state_from_vmc(v, a.walkers, replicate=True)
d.run_block(16, read=False)
eng.section_profile(reset=True)
d.run_block(a.steps, read=False)
prof = eng.section_profile(reset=True)
ser = d.read_series(a.steps)
table(f'DMC N={n}', prof, int(ser.num_walkers.sum()))
