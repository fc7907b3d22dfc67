"""CPU stand-in backend for the launcher test of bench.py (selected with
QMC_BENCH_BACKEND=tests._bench_standin).  TEST INFRASTRUCTURE: the population
handle is the oracle-backed look-alike of tests/_dist_worker.py and the
process group is gloo; it exists so that `bench.py --gpus 2` (rank spawning,
rendezvous, the sharded-DMC host logic and the JSON line) can be exercised
without a GPU.  bench.py never selects it by itself."""
import os

import numpy as np
import torch

from bench import box_spec
from oracle import qmc_oracle as orc
from phd_qmclib_amd.dist import DistributedDmc
from tests._dist_worker import OracleShard


class Backend:
    dist_backend = 'gloo'
    has_vmc = False
    name = 'cpu-standin (oracle + gloo; tests only)'
    data = 'cpu-standin'

    def __init__(self, local_rank):
        # (a rank that dies before it joins the process group: launcher test)
        if os.environ.get('QMC_STANDIN_FAIL_RANK') == str(local_rank):
            raise SystemExit(7)
        self.local_rank = local_rank
        self.device = torch.device('cpu')

    def sync(self):
        pass

    def sharded_population(self, n, start, cap, global_target, rank, world,
                           equil, rebalance_every, solo=False):
        model = orc.model_from_cfc(box_spec(n).cfc_spec)
        rng = np.random.RandomState(100 + rank)
        pos = n * rng.random_sample((start, n))
        shard = OracleShard(orc, model, pos, 6.25e-4, cap, global_target, 0.5,
                            seed=1, slot0=rank << 26)
        dd = DistributedDmc(shard, n, 'cpu', rebalance_every=rebalance_every,
                            solo=solo)
        return shard, dd, None, start
