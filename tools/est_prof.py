"""One DMC configuration with estimators for rocprofv3 --kernel-trace (dev tool)."""
import os
import sys
from math import pi

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phd_qmclib_amd.engine import DmcEnsemble, ModelEngine  # noqa
from phd_qmclib_amd.mrbp_qmc import Spec  # noqa

n, W = 64, 1 << 17
modes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
spec = Spec(lattice_depth=5 * pi ** 2, lattice_ratio=1, interaction_strength=2,
            boson_number=n, supercell_size=n, tbf_contact_cutoff=0.25 * n)
eng = ModelEngine(spec.cfc_spec, device=0)
pos = n * np.random.RandomState(1).random_sample((W, n))
maxw = ((W * 512 // 480) + 255) // 256 * 256
d = DmcEnsemble(eng, 6.25e-4, maxw, W, 0.5, rng_seed=1)
d.set_state(pos)
d.set_estimators(num_modes=modes, ssf_pure=True, ssf_pfw=6, num_bins=128)
d.run_block_est(12, True)
eng.sync()
print('done')
