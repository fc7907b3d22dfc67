"""Is the first launch after an idle GPU slow?  (development probe)"""
import os, sys, time
from math import pi
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), 'tools'))
from phd_qmclib_amd.engine import ModelEngine, VmcEnsemble
from phd_qmclib_amd.mrbp_qmc import Spec
from _stationary import replicate, seed_configurations
n, W = 64, 1 << 20
spec = Spec(lattice_depth=5 * pi ** 2, lattice_ratio=1, interaction_strength=2,
            boson_number=n, supercell_size=n, tbf_contact_cutoff=0.25 * n)
eng = ModelEngine(spec.cfc_spec, device=0)
v = VmcEnsemble(eng, W, 0.125, rng_seed=1)
v.set_state(replicate(seed_configurations(eng, spec, n), W))
v.run_block(200, sums=False); eng.sync()
def timed(tag, pre):
    v.run_block(5, sums=False); eng.sync()
    pre()
    # each launch alone: profile_begin/end around single steps is too slow;
    # time groups: first 2 launches, next 2, next 16
    out = []
    for k in (1, 1, 2, 16):
        eng.timer_start(); v.run_block(k, sums=False); ms = eng.timer_stop()
        out.append(ms / k)
    print(f'{tag:28s} first {out[0]:.4f}  second {out[1]:.4f}  next2 {out[2]:.4f}  next16 {out[3]:.4f} ms/launch', flush=True)
for rep in range(2):
    timed('back to back', lambda: None)
    timed('idle 1 ms', lambda: time.sleep(0.001))
    timed('idle 5 ms', lambda: time.sleep(0.005))
    timed('idle 50 ms', lambda: time.sleep(0.05))
    timed('idle 500 ms', lambda: time.sleep(0.5))
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29531')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
for rep in range(2):
    timed('after process group init', lambda: None)
    timed('dist.barrier + sync', lambda: (dist.barrier(), torch.cuda.synchronize()))
    timed('all_reduce 2 doubles + sync', lambda: (dist.all_reduce(torch.zeros(2, dtype=torch.float64, device='cuda:0')), torch.cuda.synchronize()))
dist.destroy_process_group()
