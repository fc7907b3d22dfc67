"""Executed instructions per kernel section (development tool).

Needs the diagnostic library built with -DQMC_CUTS, in which a wavefront ENDS at
the section mark the host selects (`ModelEngine.section_cut`):

    tools/build_variant.sh cuts "-DQMC_CUTS"
    QMCWALK_LIB=$PWD/build/variants/cuts/libqmcwalk.so tools/section_counts.sh <tag> [--bosons N]

This file is the workload the shell script runs under `rocprofv3 --pmc`: an
equilibrated VMC ensemble (full kernel), then two launches cut at every mark in
execution order, then -- for DMC -- one time step per mark from a fresh
population.  It prints the order of the cuts; the script pairs it with the
per-dispatch counters: the difference between the runs cut at successive marks is
what the section between them executes, averaged over ALL wavefronts (a section
of the energy pass is reached by the accepted moves only)."""
import argparse
import json
import os
import sys
from math import pi

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phd_qmclib_amd.engine import DmcEnsemble, ModelEngine, VmcEnsemble  # noqa
from phd_qmclib_amd.mrbp_qmc import Spec  # noqa

ap = argparse.ArgumentParser()
ap.add_argument('--bosons', type=int, default=64)
ap.add_argument('--walkers', type=int, default=1 << 16)
ap.add_argument('--equil', type=int, default=300)
ap.add_argument('--start-file', default='',
                help='.npy of stationary seed configurations')
ap.add_argument('--plan', default='')
a = ap.parse_args()
n = a.bosons
spec = Spec(lattice_depth=5 * pi ** 2, lattice_ratio=1, interaction_strength=2,
            boson_number=n, supercell_size=n, tbf_contact_cutoff=0.25 * n)
eng = ModelEngine(spec.cfc_spec, device=0)
names = eng.section_names()
ID = {nm: i for i, nm in enumerate(names)}
B = 16                                   # offset of the energy pass
inner = ['tables+onebody', 'pairs_in_lane', 'leading_short_steps',
         'rotation_loop_body', 'rotation', 'rotation_last_step', 'energy+logwf']
if n <= 64:
    inner = [s for s in inner if s not in ('pairs_in_lane', 'rotation')]
vmc_cuts = [('top', ID['top']), ('load+philox+wrap', ID['load+philox+wrap']),
            ('resort', ID['resort'])] + \
    [(s, ID[s]) for s in inner] + \
    [('metropolis+store', ID['metropolis+store']),
     ('energy_pass', ID['energy_pass'])] + \
    [(s + '@energy', ID[s] + B) for s in inner] + \
    [('store', ID['store']), ('end', ID['end'])]
dmc_cuts = [('top', ID['top']), ('load+philox+wrap', ID['load+philox+wrap']),
            ('resort', ID['resort'])] + [(s, ID[s]) for s in inner] + \
    [('weight+store', ID['weight+store']), ('end', ID['end'])]

rng = np.random.RandomState(1)
W = a.walkers
v = VmcEnsemble(eng, W, 0.25 * spec.well_width, rng_seed=1)
pos = n * rng.random_sample((W, n))
if a.start_file:                     # (tools/make_stationary.py)
    seed = np.load(a.start_file)
    pos = np.ascontiguousarray(np.tile(seed, (-(-W // len(seed)), 1))[:W])
v.set_state(pos)
done = 0
while done < a.equil:
    v.run_block(50, sums=False)
    done += 50
eng.sync()
for name, cid in vmc_cuts:
    eng.section_cut(cid)
    v.run_block(2, sums=False)
    eng.sync()
eng.section_cut(-1)
res = v.run_block(8)
acc = float(res['num_accepted'].sum() / (8 * W))
maxw = ((W * 512 // 480) + 255) // 256 * 256
for name, cid in dmc_cuts:
    eng.section_cut(-1)
    d = DmcEnsemble(eng, 6.25e-4, maxw, W, 0.5, rng_seed=1)
    d.set_state_from_vmc(v, W)
    d.run_block(2, read=False)           # two full steps: walkers, spare normals
    eng.sync()
    eng.section_cut(cid)
    d.run_block(1, read=False)
    eng.sync()
    eng.section_cut(-1)
    d.close()
plan = dict(bosons=n, walkers=W, maxw=maxw, acceptance=acc,
            vmc=[nm for nm, _ in vmc_cuts], vmc_launches_per_cut=2,
            vmc_tail_launches=8,
            dmc=[nm for nm, _ in dmc_cuts], dmc_full_steps_before_each_cut=2)
if a.plan:
    json.dump(plan, open(a.plan, 'w'))
print(json.dumps(plan))
