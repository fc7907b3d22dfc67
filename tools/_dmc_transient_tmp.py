import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from math import pi
from phd_qmclib_amd.engine import DmcEnsemble, ModelEngine, VmcEnsemble
from phd_qmclib_amd.mrbp_qmc import Spec
n=64
spec = Spec(lattice_depth=5*pi**2, lattice_ratio=1, interaction_strength=2, boson_number=n, supercell_size=n, tbf_contact_cutoff=0.25*n)
eng = ModelEngine(spec.cfc_spec, device=0)
W=1<<20
rng=np.random.RandomState(1)
v = VmcEnsemble(eng, W, 0.25*spec.well_width, rng_seed=1)
v.set_state(n*rng.random_sample((W,n)))
for _ in range(6): v.run_block(64, sums=False)
target=1<<18; maxw=((target*512//480)+255)//256*256
d = DmcEnsemble(eng, 6.25e-4, maxw, target, 0.5, rng_seed=1)
d.set_state_from_vmc(v, target, replicate=True)
for b in range(12):
    eng.profile_begin(8)
    d.run_block(8, read=False)
    nl, tot, mn, mx = eng.profile_end()
    ser = d.read_series(8)
    print(f'steps {8*b:3d}-{8*b+7:3d}: evolve {tot/nl:.4f} ms (min {mn:.4f} max {mx:.4f})  walkers {ser.num_walkers.mean():.0f}  E/N {ser.energy.sum()/ser.weight.sum()/n:.4f}', flush=True)
