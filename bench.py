#!/usr/bin/env python
"""bench.py -- walker-steps/s of the MI355X walker-propagation engine.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE
JSON line (rank 0).  With N > 1 and no RANK in the environment this process
spawns the N ranks itself (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set, before anything here touches the GPU runtime) and relays rank
0's line; under `torch.distributed.run` it is one of the ranks.

What a "step" is and which workload is the headline:

* N = 1 -- BASELINE.json configs[1]: mrbp_qmc VMC, N = 64 bosons, 2^20
  independent chains; a step is one all-particle Metropolis step of every
  chain including the local energy.  `extra.dmc` is configs[2] (DMC, N = 64,
  2^18 walkers) and `extra.c4_dmc_sharded` the multi-GPU workload run on this
  one GPU (the strong-scaling reference point).
* N > 1 -- BASELINE.json configs[3]: mrbp_qmc DMC, N = 128 bosons, ONE
  population of 2^22 walkers (global target) sharded over the ranks
  (`dist.DistributedDmc`): per time step every rank branches and propagates
  its walkers, a 16-byte RCCL all-reduce gives the global (E_t, W_t) for the
  E_ref feedback, and the ranks level their populations with point-to-point
  walker transfers.  The ranks start +-3 % off their share and the population
  is levelled by FORCED rebalances (one in the warm-up, one in the timed
  region), so that the run moves real walker records over RCCL; walker
  conservation and bit-identical E_ref on all ranks are asserted and
  `extra.phases` times the all-reduce, the host enqueue, the rebalances and the
  evolve kernel.  Total work is fixed as N grows: `"scaling": "strong"`.
  `extra.vmc_weak` is the VMC workload at 2^20 chains per GPU.

Both scaling curves from ONE run at each N: every line, at every --gpus,
carries `extra.curves = {"vmc_n64_weak": <walker-steps/s>, "dmc_n128_strong":
<walker-steps/s>}` under the same keys -- the two workloads are never mixed in
one curve, whatever `value` is at that N -- and `extra.strong_scaling` /
`extra.weak_scaling` = {ref_1gpu, speedup, efficiency}.  At N > 1 the 1-GPU
reference points are measured IN THE SAME RUN on rank 0's GPU (the full
population of configs[3] / one rank's VMC share, through the same code, while
the other ranks wait at a barrier), so that an efficiency never stitches two
boxes together; at N = 1 they are the run itself.

The ensembles are timed in the STATIONARY state of the chain, the state a
production run spends its time in.  That state is far from a random start: E/N
is 15.73 after 320 Metropolis steps from uniformly random positions, 15.43
after 10 000 and 15.395 +- 0.002 from 40 000 on, the acceptance ratio falls
from 0.47 to 0.450 on the way, and the step kernel is 5-7 % FASTER on the
stationary ensemble (fewer accepted moves, each of which costs an energy pass,
and fewer rotation steps with both pair classes in one wavefront).  Rounds 1-4
timed 320 steps after a random start; a seed ensemble of chains / 64 chains
now takes `--pre-equil` (30 000) steps from one particle per lattice well --
one second of GPU time at that size -- every chain starts from one of its
configurations and `--equil` (320) steps with the chains' own random streams
take the copies apart (`HipBackend.equilibrated_vmc`; the CPU baseline starts
from the same seed configurations).  The result windows (energy per particle,
acceptance: the stationary values) are asserted in here, and the line carries
both; with `--unrelaxed`, `extra.vmc_unrelaxed_start` is the same kernel on the
old ensemble in the same run.  Inputs are resident in HBM when a timed region starts.  `roofline` is the HBM view the metric contract
asks for: algorithmic bytes of SURVEY.md 8(d) over the dominant kernel's own
duration, measured with HIP events on the stream it is launched on; the path
is fp64-VALU bound, so `extra.valu` gives the pair-evaluation rate as well.
`cpu_baseline` (N = 1, rank 0) times the CPU oracle -- the C restatement of
the reference algorithm, OpenMP over chains, rebuilt here with -O3
-march=native -- on bounded samples of the same workload: on every core of the
affinity mask and on 16 threads (`runs`; `value` is the better), and on
BASELINE configs[0] (`c1`).  `"backend"` names the engine that produced the
line ("hip"; the CPU stand-ins of tests/ say so).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from math import pi

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
TRAFFIC_JSON = os.path.join(ROOT, 'profiles', 'traffic.json')

# result windows of the equilibrated "mrbp_qmc box" (V0 = 5 E_R, r = 1, g = 2,
# unit filling, rm = L / 4): E/N and acceptance of the trial state and the
# DMC mixed estimate, N = 64 ... 128 (tests/test_gpu_sampling.py pins them
# against the oracle; here they guard the benchmark against timing garbage)
VMC_E_WINDOW = (15.30, 15.50)       # stationary: 15.395 (15.73 after 320 steps
                                    # from a uniform random start, 15.40 after 40 000)
VMC_ACC_WINDOW = (0.43, 0.47)       # stationary: 0.450
DMC_E_WINDOW = (15.20, 15.90)       # relaxing from the VMC value towards 15.46


def box_spec(n):
    from phd_qmclib_amd.mrbp_qmc import Spec
    return Spec(lattice_depth=5 * pi ** 2, lattice_ratio=1,
                interaction_strength=2, boson_number=n, supercell_size=n,
                tbf_contact_cutoff=0.25 * n)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=64)
    ap.add_argument('--warmup', type=int, default=16)
    ap.add_argument('--equil', type=int, default=320,
                    help='untimed steps of the full ensemble before --warmup '
                         '(they take the copies of the seed configurations '
                         'apart)')
    ap.add_argument('--pre-equil', type=int, default=30000,
                    help='steps of the seed ensemble (chains / 64 chains from a '
                         'one-particle-per-well start) that brings the '
                         'configurations to the stationary state')
    ap.add_argument('--block', type=int, default=16,
                    help='Metropolis steps enqueued per block call')
    ap.add_argument('--bosons', type=int, default=64)
    ap.add_argument('--chains', type=int, default=1 << 20,
                    help='VMC chains per GPU')
    ap.add_argument('--dmc-walkers', type=int, default=1 << 18)
    ap.add_argument('--c4-bosons', type=int, default=128)
    ap.add_argument('--c4-walkers', type=int, default=1 << 22,
                    help='GLOBAL target population of the sharded DMC run')
    ap.add_argument('--rebalance-every', type=int, default=16)
    ap.add_argument('--no-dmc', action='store_true')
    ap.add_argument('--no-c4', action='store_true',
                    help='skip the sharded-DMC workload at --gpus 1')
    ap.add_argument('--no-vmc-extra', action='store_true',
                    help='skip the weak-scaled VMC extra at --gpus > 1')
    ap.add_argument('--no-scaling-ref', action='store_true',
                    help='--gpus > 1: skip the same-run 1-GPU reference points '
                         '(rank 0 alone: the whole DMC population, one VMC '
                         'share) behind extra.strong_scaling / weak_scaling')
    ap.add_argument('--unrelaxed', action='store_true',
                    help='also time the headline kernel on the UNRELAXED ensemble '
                         'rounds 1-3 timed (--equil steps after a uniform random '
                         'start) as an extra line (never the headline; off by '
                         'default: under a profiler its launches would share '
                         'the timed instantiation\'s row of the summary)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-checks', action='store_true',
                    help='do not assert the energy / acceptance windows '
                         '(other models or sizes than the benchmark box)')
    ap.add_argument('--fp32', action='store_true',
                    help='also time the reduced-precision (fp32 pair loop) '
                         'variant as an extra line (never the headline)')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--start-skew', type=float, default=0.03,
                    help='--gpus > 1: even ranks start with this fraction more '
                         'walkers than their share, odd ranks with as many '
                         'fewer, so that the population rebalance has real '
                         'transfers to make')
    ap.add_argument('--launch-timeout', type=float, default=3000.0,
                    help='seconds after which the launcher stops its ranks')
    return ap.parse_args(argv)


# ---------------------------------------------------------------- launcher --
def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """Spawn one child per GPU and relay rank 0's JSON line.  Nothing in this
    process has touched the GPU runtime (no torch.cuda, no HIP call), and no
    process is replaced: the children are ordinary subprocesses."""
    import threading
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get(
                       'HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen(
            [sys.executable, os.path.abspath(__file__)] + list(argv),
            env=env, stdout=subprocess.PIPE if r == 0 else None))
    # rank 0's stdout is drained while it runs (a rank that has written more
    # than a pipe buffer would otherwise block in write() for ever)
    chunks = []
    reader = threading.Thread(
        target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()

    # a rank that dies leaves the others waiting in a collective: stop them
    # (the processes started here, by handle) instead of waiting for the
    # process group's timeout; the whole run has a deadline as well
    rc = 0
    deadline = time.time() + args.launch_timeout
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            stop_all()
            break
        if all(c is not None for c in codes):
            break
        if time.time() > deadline:
            sys.stderr.write(f'bench.py: the ranks did not finish within '
                             f'{args.launch_timeout:.0f} s; stopping them\n')
            rc = 124
            stop_all()
            break
        time.sleep(0.2)
    reader.join(timeout=30)
    out = b''.join(chunks)
    if rc == 0:
        sys.stdout.write(out.decode())
        sys.stdout.flush()
    elif rc != 124:
        sys.stderr.write(f'bench.py: a rank exited with status {rc}\n')
    return rc


# ----------------------------------------------------------------- backend --
class HipBackend:
    """The product path: HIP engine + RCCL.  (The CPU tests of the launcher
    substitute a stand-in through QMC_BENCH_BACKEND=<module>; nothing in this
    file falls back to it.)"""
    dist_backend = 'nccl'
    has_vmc = True
    name = 'hip'                 # -> "backend" of the JSON line
    data = 'synthetic'

    def __init__(self, local_rank):
        import torch
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs a GPU (the HIP engine has no '
                             'CPU path)')
        torch.cuda.set_device(local_rank)
        self.torch = torch
        self.local_rank = local_rank
        self.device = torch.device('cuda', local_rank)
        # one side stream for everything: the engine launches on it and RCCL
        # orders its collectives with it (dist.DistributedDmc checks this)
        self.stream = torch.cuda.Stream(device=self.device)
        torch.cuda.set_stream(self.stream)
        self._engines = {}

    def engine(self, n, fast_math=False):
        from phd_qmclib_amd.engine import ModelEngine
        key = (n, bool(fast_math))
        if key not in self._engines:
            kw = dict(fast_math=True) if fast_math else {}
            self._engines[key] = ModelEngine(
                box_spec(n).cfc_spec, device=self.local_rank,
                stream=self.stream.cuda_stream, **kw)
        return self._engines[key]

    def sync(self):
        self.torch.cuda.synchronize()

    # Metropolis steps of the seed ensemble (run_rank sets it from --pre-equil)
    pre_equil = 30000
    seed_confs = {}

    def equilibrated_vmc(self, n, chains, chain0, equil, seed_rank,
                         fast_math=False, unrelaxed=False):
        """A VMC ensemble of `chains` chains of the N = n box IN EQUILIBRIUM.

        The chain relaxes slowly: from a uniform random start E/N is 15.66
        after 1000 steps, 15.43 after 10 000 and 15.395 -- its stationary
        value -- after 40 000, the acceptance ratio falls from 0.48 to 0.450
        on the way, and the step kernel is 5 % slower on the unrelaxed
        ensemble than on the stationary one (more accepted moves, each with its
        energy pass; `profiles/r04_ab_variants.txt` section 12).  A production
        run spends its time in the stationary state, so that is what is timed:
        a SEED ensemble of chains / 64 chains starts with one particle per
        lattice well and takes `pre_equil` steps (30 000: a second of GPU time
        at this size; E/N and acceptance are stationary to 10^-3 after
        20 000), every chain of the full ensemble starts from one of its
        configurations, and `equil` further steps with the chains' own random
        streams take the copies apart."""
        import numpy as np
        from phd_qmclib_amd.engine import VmcEnsemble
        spec = box_spec(n)
        eng = self.engine(n, fast_math)
        rng = np.random.RandomState(1000 + seed_rank)
        spread = 0.25 * spec.well_width
        if unrelaxed:
            # what rounds 1-3 timed: `equil` steps after a uniform random start
            v = VmcEnsemble(eng, chains, spread, rng_seed=1, chain0=chain0)
            pos = np.empty((chains, n))
            for lo in range(0, chains, 1 << 16):
                hi = min(chains, lo + (1 << 16))
                pos[lo:hi] = spec.supercell_size * rng.random_sample(
                    (hi - lo, n))
            v.set_state(pos)
            del pos
            done = 0
            while done < equil:
                b = min(64, equil - done)
                v.run_block(b, sums=False)
                done += b
            return v
        seeds = min(chains, max(2048, chains // 64))
        # the wells of the lattice are [i, i + well_width), i = 0 ... N - 1
        pos = (np.arange(n)[None, :] + 0.5 * spec.well_width +
               0.6 * spec.well_width * (rng.random_sample((seeds, n)) - 0.5))
        v = VmcEnsemble(eng, seeds, spread, rng_seed=2, chain0=chain0)
        v.set_state(pos)
        done = 0
        while done < self.pre_equil:
            b = min(128, self.pre_equil - done)
            # (through the SERIES instantiation of the step kernel -- the same
            # trajectories bit for bit, tests/test_gpu_scale.py -- so that the
            # per-kernel summary of a profiled run keeps the timed
            # instantiation's launches apart from these 30 000 small ones)
            v.run_block(b, sums=False, series='stat')
            done += b
        pos = v.get_state()[0]
        v.close()
        if not fast_math:
            self.seed_confs[n] = pos       # (cpu_baseline starts from them too)
        reps = -(-chains // seeds)
        v = VmcEnsemble(eng, chains, spread, rng_seed=1, chain0=chain0)
        # (2^20 x 64 doubles: 512 MiB on the host)
        v.set_state(np.tile(pos, (reps, 1))[:chains])
        del pos
        done = 0
        while done < equil:
            b = min(64, equil - done)
            v.run_block(b, sums=False)
            done += b
        return v

    def sharded_population(self, n, start, cap, global_target, rank, world,
                           equil, rebalance_every, solo=False):
        """This rank's share of ONE DMC population (external reduce), `start`
        walkers to begin with, and its DistributedDmc driver; the walkers
        start from equilibrated VMC configurations, reused cyclically when
        there are fewer chains.  `solo`: the whole population on this rank,
        no collective whatever the process group (the 1-GPU reference)."""
        import torch.distributed as dist
        from phd_qmclib_amd.dist import DistributedDmc
        from phd_qmclib_amd.engine import DmcEnsemble
        torch = self.torch
        eng = self.engine(n)
        chains = min(start, 1 << 17)
        v = self.equilibrated_vmc(n, chains, rank * chains, equil, rank)
        # (Philox slot range of a rank: 2^26 slots, whatever the rank count)
        d = DmcEnsemble(eng, 6.25e-4, cap, global_target, 0.5, rng_seed=1,
                        slot0=rank << 26, external_reduce=True)
        d.set_state_from_vmc(v, start, replicate=True)
        # every rank must start from the same E_ref: the global mean energy
        er = torch.tensor([d.get_scalars()[2]], dtype=torch.float64,
                          device=self.device)
        if world > 1:
            dist.all_reduce(er)
        d.set_state_from_vmc(v, start, ref_energy=float(er.item()) / world,
                             replicate=True)
        v.close()
        dd = DistributedDmc(d, n, self.device,
                            rebalance_every=rebalance_every, solo=solo)
        return d, dd, eng, chains


def load_backend(local_rank):
    name = os.environ.get('QMC_BENCH_BACKEND')
    if name:                     # tests/ only: launcher + host logic on CPU
        import importlib
        return importlib.import_module(name).Backend(local_rank)
    return HipBackend(local_rank)


def check_window(what, value, window, args):
    if args.no_checks:
        return
    lo, hi = window
    if not (lo <= value <= hi):
        raise SystemExit(f'bench.py: {what} = {value:.5f} outside the '
                         f'expected window [{lo}, {hi}] -- refusing to '
                         f'report a throughput for a wrong result')


def loaded_kernel_hash():
    """`qmc_source_hash()` of the library this process runs (None for the CPU
    stand-ins of tests/, which load no kernels)."""
    try:
        from phd_qmclib_amd import _lib
        return _lib.source_hash()
    except Exception:
        return None


def load_traffic(n, kernel, path=None, loaded=None):
    """Measured HBM bytes / VALU instructions per unit of `kernel`
    ('vmc_step_kernel' / 'dmc_evolve_kernel') at N = n from the committed
    rocprofv3 PMC passes (tools/profile.sh -> tools/make_traffic.py ->
    profiles/traffic.json) -- IF they were taken on the kernels this process
    runs: the file records `qmc_source_hash()` of the measured library, and a
    figure whose hash differs from the loaded library's is stale and not
    reported.  -> (dict of figures without the kernel prefix, note or None)."""
    path = TRAFFIC_JSON if path is None else path
    try:
        with open(path) as fp:
            ent = json.load(fp).get(f'N{n}', {})
    except (OSError, ValueError):
        return {}, f'no traffic record ({os.path.basename(path)} unreadable)'
    pre = kernel + '_'
    ent = {k[len(pre):]: v for k, v in ent.items() if k.startswith(pre)}
    if not ent:
        return {}, f'no traffic record for {kernel} at N={n}'
    loaded = loaded_kernel_hash() if loaded is None else loaded
    measured = ent.get('kernel_source_sha')
    if measured is None or loaded is None or measured != loaded:
        return {}, (f'profiles/traffic.json was measured on kernels {measured} '
                    f'but this run loaded {loaded}: stale, not reported -- '
                    f'rerun tools/profile.sh + tools/make_traffic.py')
    return ent, None


# ---------------------------------------------------------------- VMC leg ---
def bench_vmc(be, args, rank, world, use_pg, n, W, fast_math=False,
              unrelaxed=False):
    """Timed VMC run of this rank's W chains -> dict of raw measurements."""
    import torch
    import torch.distributed as dist
    eng = be.engine(n, fast_math)
    if unrelaxed:
        vmc = be.equilibrated_vmc(n, W, rank * W, args.equil, rank, fast_math,
                                  unrelaxed=True)
    else:
        vmc = be.equilibrated_vmc(n, W, rank * W, args.equil, rank, fast_math)

    def barrier():
        if use_pg:
            dist.barrier()
        be.sync()

    def run_steps(k):
        done = 0
        while done < k:
            b = min(args.block, k - done)
            vmc.run_block(b, sums=False)
            done += b
        return k

    # (ranks meet BEFORE the warm-up, so that they reach the barrier in front
    # of the timed region together: a rank that waits there idles, and a GPU
    # that idled for milliseconds starts the timed region 15-25 % slow --
    # `profiles/r04_ab_variants.txt` section 23)
    barrier()
    run_steps(args.warmup)
    # (device counter of walkers that leave the sorted-row pair sums: read
    # over the timed region -- which code did the timed kernel run?)
    gp = getattr(eng, 'general_path_walkers', None)
    if gp is not None:
        gp(reset=True)
    barrier()
    eng.timer_start()
    t0 = time.perf_counter()
    launches = run_steps(args.steps)
    kernel_ms = eng.timer_stop()          # HIP events on the launch stream
    barrier()
    dt = time.perf_counter() - t0
    gp_timed = None if gp is None else int(gp(reset=True))
    if use_pg:
        tmax = torch.tensor([dt], dtype=torch.float64, device=be.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    # per-launch spread: every launch bracketed by its own event pair
    eng.profile_begin(args.block * 4)
    vmc.run_block(args.block * 4, sums=False)
    nl, tot_ms, min_ms, max_ms = eng.profile_end()
    # block estimators of one more block (global reduction over ranks)
    res = vmc.run_block(args.block, sums=True)
    tot = torch.tensor([res['sum_energy'].sum(),
                        float(res['num_accepted'].sum()),
                        float(W * args.block)], dtype=torch.float64,
                       device=be.device)
    if use_pg:
        dist.all_reduce(tot)
    tot = tot.cpu().numpy()
    return dict(vmc=vmc, eng=eng, dt=dt, kernel_ms=kernel_ms,
                launches=launches, general_path_walkers=gp_timed,
                launch_ms_avg_isolated=tot_ms / max(nl, 1),
                launch_ms_min=min_ms, launch_ms_max=max_ms,
                energy_per_particle=float(tot[0] / tot[2] / n),
                accept_rate=float(tot[1] / tot[2]))


def vmc_line(args, m, n, W, world):
    """The headline JSON object of the VMC workload from measurements `m`."""
    value = world * W * args.steps / m['dt']
    b_vmc = 16 * n + 32                      # SURVEY.md 8(d), bytes/chain-step
    launch_ms = m['kernel_ms'] / m['launches']
    achieved = W * b_vmc / (launch_ms * 1e-3) / 1e9
    pairs = n * (n - 1) // 2
    tj, traffic_note = load_traffic(n, 'vmc_step_kernel')
    traffic = tj.get('bytes_per_chain_step')
    valu = tj.get('valu_instr_per_chain_step')
    return {
        'metric': 'walker-steps/sec',
        'value': value,
        'unit': 'walker-steps/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': m['dt'] / args.steps * 1e3,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': {
            'workload': f'mrbp_qmc VMC, N={n} bosons, {W} chains per GPU, '
                        f'move_spread=0.25*well_width, energy on accepted '
                        f'moves, stationary ensemble (seed chains: '
                        f'{args.pre_equil} steps from one particle per well; '
                        f'copies + {args.equil} steps)',
            'bosons': n, 'chains_per_gpu': W, 'steps_per_launch': 1,
            'parallelism': f'chains sharded over {world} GPU(s), no '
                           f'data-path collective',
        },
        'roofline': {
            'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
            'traffic': None if traffic is None else traffic * W,
            'traffic_note': traffic_note or
            'rocprofv3 PMC passes of these kernels (same qmc_source_hash), '
            'profiles/traffic.json',
            'kernel': 'vmc_step_kernel', 'launch_ms': launch_ms,
            'bytes_per_unit': b_vmc, 'units_per_launch': W,
        },
        'extra': {
            'valu': {
                'pair_evals_per_s': W * args.steps * pairs /
                (m['kernel_ms'] * 1e-3),
                # the ceiling that binds: wave64 VALU instructions issued
                # (SQ_INSTS_VALU / SQ_WAVES of the rocprofv3 PMC pass) against
                # 1 instruction per 4 cycles per SIMD, 1024 SIMDs at 2.4 GHz
                'instr_per_chain_step': valu,
                'issue_frac': None if valu is None else
                (W * valu / (launch_ms * 1e-3)) / (1024 * 2.4e9 / 4),
                'note': 'the path is fp64-VALU bound (SURVEY.md 8d); unique '
                        'pairs N(N-1)/2 per chain-step'},
            'launch_ms_isolated_avg': m['launch_ms_avg_isolated'],
            'launch_ms_min': m['launch_ms_min'],
            'launch_ms_max': m['launch_ms_max'],
            'vmc_energy_per_particle': m['energy_per_particle'],
            'vmc_accept_rate': m['accept_rate'],
        },
    }


# ---------------------------------------------------------------- DMC legs --
def bench_dmc_single(be, args, n, vmc, target):
    """configs[2]: one DMC population on this GPU, started device to device
    from the equilibrated VMC chains."""
    from phd_qmclib_amd.engine import DmcEnsemble
    eng = be.engine(n)
    maxw = ((target * 512 // 480) + 255) // 256 * 256
    d = DmcEnsemble(eng, 6.25e-4, maxw, target, 0.5, rng_seed=1)
    d.set_state_from_vmc(vmc, target, replicate=True)
    # (at least 64 steps, 30 ms: the population was set up just now with the
    # GPU idle, and the chip needs that long to come back -- section 23)
    d.run_block(max(args.warmup, 64), read=False)
    eng.sync()
    # three timed regions of --steps steps, the median reported: 20 steps are
    # 10 ms of wall clock, and one descheduling of the enqueueing thread in
    # there has been seen to cost 3.8 ms (profiles/README.md, r04_bench.json)
    runs = []
    for _ in range(3):
        eng.profile_begin(args.steps)
        t0 = time.perf_counter()
        d.run_block(args.steps, read=False)
        nl, evolve_ms, _, _ = eng.profile_end()       # synchronises
        ddt = time.perf_counter() - t0
        ser = d.read_series(args.steps)
        runs.append((ddt, nl, evolve_ms, ser))
    wall_ms = [r[0] / args.steps * 1e3 for r in runs]
    ddt, nl, evolve_ms, ser = sorted(runs, key=lambda r: r[0])[1]
    nws = float(ser.num_walkers.sum())
    b_dmc = 32 * n + 40 + 16
    e_per = float(ser.energy.sum() / ser.weight.sum() / n)
    d.close()
    check_window('DMC energy per particle', e_per, DMC_E_WINDOW, args)
    return {
        'workload': f'mrbp_qmc DMC, N={n}, target {target} / max {maxw} '
                    f'walkers, dt=6.25e-4, from equilibrated VMC chains',
        'walker_steps_per_s': nws / ddt,
        'ms_per_step': ddt / args.steps * 1e3,
        'ms_per_step_runs': wall_ms,       # (the median is reported)
        'evolve_kernel_ms': evolve_ms / max(nl, 1),
        'hbm_achieved_GBs': nws * b_dmc / (evolve_ms * 1e-3) / 1e9,
        'hbm_frac': nws * b_dmc / (evolve_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        'mean_walkers': nws / args.steps,
        'energy_per_particle': e_per,
    }


def bench_dmc_sharded(be, args, rank, world, use_pg, solo=False):
    """configs[3]: ONE population of --c4-walkers (global target) sharded over
    the ranks.  -> dict (identical on every rank).

    At world > 1 the ranks start deliberately unequal (+-`--start-skew` of
    their share) and the population is levelled by FORCED rebalances -- one in
    the warm-up, one in the middle of the timed region -- so that the run
    exercises the whole RCCL data path (counts all-gather, batched point-to-
    point walker transfers, import at the tail) and not only the 16-byte
    all-reduce; walker conservation across every transfer and identical E_ref
    on all ranks in every step are asserted."""
    import numpy as np
    import torch
    import torch.distributed as dist
    n = args.c4_bosons
    target = args.c4_walkers
    per_rank = target // world
    cap = ((per_rank * 512 // 480) + 255) // 256 * 256
    start = per_rank
    if world > 1 and args.start_skew > 0 and rank < world - world % 2:
        delta = int(per_rank * args.start_skew)
        start = per_rank + (delta if rank % 2 == 0 else -delta)
    kw = dict(solo=True) if solo else {}
    d, dd, eng, chains = be.sharded_population(
        n, start, cap, target, rank, world, args.equil, args.rebalance_every,
        **kw)

    def barrier():
        if use_pg:
            dist.barrier()
        be.sync()

    checks = {}

    def forced_rebalance(tag):
        """-> walkers this rank sent or received; checks conservation."""
        before = dd.global_counts()
        moved = dd.rebalance(force=True)
        after = dd.global_counts()
        if sum(before) != sum(after):
            raise SystemExit(f'bench.py: {tag} rebalance lost walkers: '
                             f'{before} -> {after}')
        if max(after) - min(after) > 1:
            raise SystemExit(f'bench.py: {tag} rebalance left {after}')
        checks[tag] = dict(counts_before=before, counts_after=after)
        return moved

    warm = max(args.warmup, 1)
    dd.run_block(1)
    moved_warm = forced_rebalance('warmup') if world > 1 else 0
    if warm > 1:
        dd.run_block(warm - 1)
    barrier()
    dd.enable_phase_timing(args.steps + 8)
    moved0 = dd.walkers_moved
    if eng is not None:
        eng.profile_begin(args.steps)
    t0 = time.perf_counter()
    h = args.steps // 2
    sers = []
    if h:
        sers.append(dd.run_block(h))
    moved_timed = forced_rebalance('timed') if world > 1 else 0
    sers.append(dd.run_block(args.steps - h))
    barrier()
    ddt = time.perf_counter() - t0
    phases = dd.phase_report()
    nl, evolve_ms = 0, 0.0
    if eng is not None:
        nl, evolve_ms, _, _ = eng.profile_end()
    cat = lambda name, col: np.concatenate(        # noqa: E731
        [np.asarray(_series_field(x, name, col), dtype=np.float64)
         for x in sers])
    nw_local = float(np.sum(cat('num_walkers', 2)))
    e_glob, w_glob = cat('energy', 0), cat('weight', 1)
    e_ref = torch.tensor(cat('ref_energy', 3), dtype=torch.float64,
                         device=be.device)
    loc = torch.tensor([nw_local, ddt, evolve_ms,
                        float(dd.walkers_moved - moved0),
                        float(dd.rebalances), float(moved_warm),
                        phases['allreduce_us_per_step'],
                        phases['host_enqueue_us_per_step'],
                        phases['rebalance_ms_total']],
                       dtype=torch.float64, device=be.device)
    tmx = loc.clone()
    if use_pg:
        dist.all_reduce(loc)                         # sums
        dist.all_reduce(tmx, op=dist.ReduceOp.MAX)   # maxima
        # the population-control feedback must be the same number everywhere
        hi, lo = e_ref.clone(), e_ref.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        if float((hi - lo).abs().max().item()) != 0.0:
            raise SystemExit('bench.py: the ranks disagree on E_ref')
    nws = float(loc[0].item())
    tmax = float(tmx[1].item())
    evolve_max = float(tmx[2].item())
    moved_all = float(loc[3].item())
    # the warm-up rebalance levels the deliberate skew: it must have moved
    # walkers (the timed one moves whatever the populations drifted apart by
    # since -- reported, and zero only in a run of one or two steps)
    if world > 1 and args.start_skew > 0 and float(loc[5].item()) <= 0:
        raise SystemExit('bench.py: the forced rebalance moved no walker; '
                         'the RCCL transfer path was not exercised')
    e_per = float(np.sum(e_glob) / np.sum(w_glob) / n)
    check_window('sharded DMC energy per particle', e_per, DMC_E_WINDOW, args)
    b_dmc = 32 * n + 40 + 16
    out = {
        'workload': f'mrbp_qmc DMC, N={n} bosons, ONE population of target '
                    f'{target} walkers sharded over {world} rank(s) '
                    f'({per_rank} per rank, cap {cap}), dt=6.25e-4, 16-byte '
                    f'all-reduce of (E_t, W_t) per step, rebalance check '
                    f'every {args.rebalance_every} steps + one forced '
                    f'rebalance in the warm-up and one in the timed region; '
                    f'ranks start +-{args.start_skew:.0%} off their share; '
                    f'walkers from {chains} equilibrated VMC chains per rank',
        'walker_steps_per_s': nws / tmax,
        'ms_per_step': tmax / args.steps * 1e3,
        'walker_steps': nws,
        'bosons': n, 'global_target': target, 'per_rank': per_rank,
        'mean_walkers': nws / args.steps,
        'energy_per_particle': e_per,
        # sends + receives, summed over the ranks (each walker counts twice)
        'walkers_moved_all_ranks': moved_all,
        'walkers_moved_warmup_all_ranks': float(loc[5].item()),
        'rebalances_max': float(tmx[4].item()),
        'ref_energy_identical_on_all_ranks': True,
        'rebalance_checks': checks,
        'bytes_per_unit': b_dmc,
        'phases': {
            'steps': args.steps,
            'allreduce_us_per_step_mean_over_ranks': float(loc[6].item()) / world,
            'allreduce_us_per_step_max_over_ranks': float(tmx[6].item()),
            'allreduce_timed_with': phases['allreduce_timed_with'],
            'host_enqueue_us_per_step_max_over_ranks': float(tmx[7].item()),
            'rebalance_ms_total_max_over_ranks': float(tmx[8].item()),
            'rebalance_calls': phases['rebalance_calls'],
            'note': 'rebalance time includes its two counts all-gathers '
                    '(host-synchronising); everything else is enqueued '
                    'without a host sync',
        },
    }
    if nl:
        # the slowest rank's evolve kernel: walkers per launch on that rank ~
        # nws / world / steps
        launch_ms = evolve_max / nl
        units = nws / world / args.steps
        out.update(evolve_kernel_ms=launch_ms,
                   hbm_achieved_GBs=units * b_dmc / (launch_ms * 1e-3) / 1e9,
                   units_per_launch=units)
        out['phases']['evolve_kernel_ms_per_step'] = launch_ms
    if hasattr(d, 'close'):
        d.close()
    return out


def _series_field(ser, name, col):
    if hasattr(ser, name):
        return getattr(ser, name)
    return ser[:, col]


def sharded_line(args, s, world):
    """Headline JSON object of the sharded DMC workload."""
    n = args.c4_bosons
    achieved = s.get('hbm_achieved_GBs')
    tj, traffic_note = load_traffic(n, 'dmc_evolve_kernel')
    traffic = tj.get('bytes_per_walker_step')
    return {
        'metric': 'walker-steps/sec',
        'value': s['walker_steps_per_s'],
        'unit': 'walker-steps/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': s['ms_per_step'],
        'higher_is_better': True,
        'scaling': 'strong',
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': {
            'workload': s['workload'],
            'bosons': n, 'global_target_walkers': s['global_target'],
            'walkers_per_gpu': s['per_rank'],
            'parallelism': f'walkers sharded over {world} GPU(s); per step '
                           f'one 16-byte RCCL all-reduce; p2p population '
                           f'rebalance',
        },
        'roofline': {
            'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s',
            'frac': None if achieved is None else achieved / HBM_PEAK_GBS,
            'traffic': None if traffic is None or 'units_per_launch' not in s
            else traffic * s['units_per_launch'],
            'traffic_note': traffic_note or
            'rocprofv3 PMC passes of these kernels (same qmc_source_hash), '
            'profiles/traffic.json',
            'kernel': 'dmc_evolve_kernel',
            'launch_ms': s.get('evolve_kernel_ms'),
            'bytes_per_unit': s['bytes_per_unit'],
            'units_per_launch': s.get('units_per_launch'),
        },
        'extra': {
            'dmc_energy_per_particle': s['energy_per_particle'],
            'mean_walkers': s['mean_walkers'],
            'walkers_moved_all_ranks': s['walkers_moved_all_ranks'],
            'walkers_moved_warmup_all_ranks':
                s['walkers_moved_warmup_all_ranks'],
            'rebalances_max': s['rebalances_max'],
            'ref_energy_identical_on_all_ranks':
                s['ref_energy_identical_on_all_ranks'],
            'rebalance_checks': s['rebalance_checks'],
            'phases': s['phases'],
        },
    }


# ------------------------------------------------------------ CPU baseline --
def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _time_oracle_vmc(orc, spec, n, move_spread, chains, threads, seconds,
                     max_steps=None, start=None):
    """Time `orc.vmc_ensemble` on `chains` chains with `threads` OpenMP
    threads for about `seconds` (or exactly `max_steps` steps if that takes
    less), from the configurations `start` (reused cyclically) or a uniform
    random start.  -> (chain-steps per second, steps run)."""
    import numpy as np
    m = orc.model_from_cfc(spec.cfc_spec)
    rng = np.random.RandomState(7)
    if start is not None:
        cpos = np.ascontiguousarray(
            np.tile(start, (-(-chains // len(start)), 1))[:chains])
    else:
        cpos = spec.supercell_size * rng.random_sample((chains, n))
    cwf = np.array([orc.wf_abs_log(m, cpos[i]) for i in range(chains)])
    cec = np.zeros(chains)
    ns = 4
    t0 = time.perf_counter()
    orc.vmc_ensemble(m, cpos, cwf, cec, move_spread, 1, ns,
                     yield_initial=True, nthreads=threads)
    probe = time.perf_counter() - t0
    ns2 = max(4, int(ns * seconds / max(probe, 1e-4)))
    if max_steps is not None:
        ns2 = min(ns2, max_steps)
    t0 = time.perf_counter()
    orc.vmc_ensemble(m, cpos, cwf, cec, move_spread, 1, ns2, step0=ns,
                     nthreads=threads)
    cdt = time.perf_counter() - t0
    return chains * ns2 / cdt, ns2


def cpu_baseline(args, spec, n, move_spread, start=None):
    """The oracle (C restatement of the reference algorithm, OpenMP over
    chains = the reference's `prange`) timed on this box's host cores, on
    bounded samples of the VMC workload.  Built here with -O3
    -march=native and libm builtins allowed (what numba's LLVM would inline;
    the committed checker build is -O2 -fno-builtin, bit-exact).  `value` uses
    every core of this process's affinity mask; `threads16` is the same on 16
    threads (a one-GPU share of the host); `c1` is BASELINE configs[0] as is
    (N = 16, 1024 chains, 16 x 512 steps) when it fits the time bound."""
    from oracle import qmc_oracle as orc
    flags = orc.build_native()          # -> flag string of the timed build
    affinity = len(os.sched_getaffinity(0))
    cores = max(1, min(orc.max_threads(), affinity))
    per = max(args.cpu_seconds / 3.0, 0.5)
    runs = {}
    for tag, th in (('all_cores', cores), ('threads16', min(16, cores))):
        if tag == 'threads16' and th == cores:
            continue
        wc = 64 * th
        rate, ns2 = _time_oracle_vmc(orc, spec, n, move_spread, wc, th, per,
                                     start=start)
        runs[tag] = {'value': rate, 'cores': th,
                     'sample': f'{wc} chains x {ns2} steps'}
    # the headline is the better of the two: on a GPU box whose host is shared
    # (a CPU quota below the affinity mask) more threads than the quota only
    # add scheduling overhead
    best = max(runs, key=lambda k: runs[k]['value'])
    out = {
        'value': runs[best]['value'], 'unit': 'walker-steps/s',
        'cores': runs[best]['cores'], 'kind': 'port',
        'sample': f"{runs[best]['sample']} of the same VMC workload (N={n}"
                  + (', from the stationary seed configurations of the GPU '
                     'run' if start is not None else '') + '), '
                  f'oracle/qmc_oracle.c with OpenMP over chains; the better '
                  f'of `runs` ({best})',
        'runs': runs,
        'cpu_model': cpu_model(),
        'compiler_flags': flags,
        'hardware_threads': os.cpu_count(),
        'affinity_cores': affinity,
        'omp_max_threads': orc.max_threads(),
    }
    # BASELINE configs[0]: the reference's own CPU-runnable case
    # (tests/mrbp_qmc/test_vmc.py:23: move_spread = 0.25 * well_width)
    s1 = box_spec(16)
    c1_steps = 16 * 512
    r1, n1 = _time_oracle_vmc(orc, s1, 16, 0.25 * s1.well_width, 1024,
                              min(cores, 1024), per, max_steps=c1_steps)
    out['c1'] = {
        'value': r1, 'unit': 'walker-steps/s', 'cores': min(cores, 1024),
        'workload': 'mrbp_qmc VMC, N=16 bosons, 1024 chains (BASELINE '
                    'configs[0])',
        'sample': f'1024 chains x {n1} of the {c1_steps} steps '
                  f'(16 blocks x 512)' + (' -- the whole case'
                                          if n1 == c1_steps else ''),
    }
    return out


# ----------------------------------------------------------------- scaling --
def scaling_entry(curve, value, ref, world, how, detail):
    """{ref_1gpu, speedup, efficiency} of one curve at this N: speedup =
    value / ref_1gpu, efficiency = speedup / N (strong scaling: the same total
    work N times faster; weak scaling: N times the work in the same time --
    `value` is the whole-job rate in both).  None when a leg was skipped."""
    if value is None or ref is None or ref <= 0:
        return None
    out = {'curve': curve, 'n_gpus': world, 'value': value, 'ref_1gpu': ref,
           'speedup': value / ref, 'efficiency': value / ref / world,
           'ref_measured': how}
    if detail:
        out['ref_detail'] = detail
    return out


# ------------------------------------------------------------------- main ---
def run_rank(args):
    # stdout carries ONE JSON line and nothing else: libraries that print to
    # file descriptor 1 (RCCL's version banner at communicator creation) go
    # to stderr for the lifetime of the rank
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE is '
                         f'{world}: launch {args.gpus} ranks (or run '
                         f'`python bench.py --gpus {args.gpus}` and let it '
                         f'spawn them)')
    import torch.distributed as dist
    be = load_backend(local_rank)
    be.pre_equil = args.pre_equil
    use_pg = world > 1 or 'RANK' in os.environ
    if use_pg:
        kw = {}
        if be.dist_backend == 'nccl':
            kw['device_id'] = be.device
        dist.init_process_group(be.dist_backend, **kw)
        world = dist.get_world_size()      # what the process group reports
        rank = dist.get_rank()
        # The first collective of a process creates the communicator (hundreds
        # of milliseconds with RCCL).  Left to the barrier in front of the first
        # timed region, that is hundreds of milliseconds of idle GPU right
        # before it -- and an MI355X that has idled for more than a few
        # milliseconds runs its next launches 15-25 % slower and needs some
        # 30 ms to come back (`profiles/r04_ab_variants.txt` section 23).
        # Create it here, long before anything is timed.
        import torch
        warm = torch.zeros(2, dtype=torch.float64, device=be.device)
        dist.all_reduce(warm)
        dist.barrier()
        be.sync()

    n, W = args.bosons, args.chains
    out = None
    # the two curves, under the same keys at every --gpus (see the docstring)
    key_v = f'vmc_n{n}_weak'
    key_d = f'dmc_n{args.c4_bosons}_strong'
    curves = {key_v: None, key_d: None}
    ref_v = ref_d = None              # 1-GPU reference points of this run
    ref_how = 'this run (N = 1)'
    ref_detail = None
    if world == 1:
        # ---- headline: VMC, configs[1] ----
        m = bench_vmc(be, args, rank, world, use_pg, n, W)
        check_window('VMC energy per particle', m['energy_per_particle'],
                     VMC_E_WINDOW, args)
        check_window('VMC acceptance', m['accept_rate'], VMC_ACC_WINDOW, args)
        out = vmc_line(args, m, n, W, world)
        curves[key_v] = ref_v = out['value']
        # walkers of the timed region evaluated by the general pair sum
        # instead of the sorted-row one (of W x steps evaluations)
        out['extra']['general_path_walkers_timed'] = m['general_path_walkers']
        if not args.no_dmc:
            out['extra']['dmc'] = bench_dmc_single(be, args, n, m['vmc'],
                                                   args.dmc_walkers)
        m['vmc'].close()
        if args.fp32:
            mf = bench_vmc(be, args, rank, world, use_pg, n, W,
                           fast_math=True)
            lf = vmc_line(args, mf, n, W, world)
            out['extra']['vmc_fp32_pair_loop'] = {
                'dtype': 'f32 pair loop (tables, sums, one-body, Metropolis '
                         'in f64)',
                'value': lf['value'], 'ms_per_step': lf['ms_per_step'],
                'energy_per_particle': mf['energy_per_particle'],
                'accept_rate': mf['accept_rate'],
                'note': 'jit_fastmath analogue of the reference '
                        '(mrbp_qmc/dmc.py:159-160); NOT the headline',
            }
            mf['vmc'].close()
        if args.unrelaxed:
            # the same kernel on the ensemble rounds 1-3 timed, in this run on
            # this box: the line says what the state of the ensemble is worth
            mu = bench_vmc(be, args, rank, world, use_pg, n, W, unrelaxed=True)
            lu = vmc_line(args, mu, n, W, world)
            out['extra']['vmc_unrelaxed_start'] = {
                'value': lu['value'], 'ms_per_step': lu['ms_per_step'],
                'energy_per_particle': mu['energy_per_particle'],
                'accept_rate': mu['accept_rate'],
                'general_path_walkers_timed': mu['general_path_walkers'],
                'note': f'the headline workload {args.equil} steps after a '
                        f'uniform random start -- the ensemble bench.py timed '
                        f'in rounds 1-3 -- instead of the stationary state: '
                        f'more accepted moves, each with its energy pass, and '
                        f'more rotation steps with both pair classes in a '
                        f'wavefront; NOT the headline',
            }
            mu['vmc'].close()
        if not args.no_c4:
            s = bench_dmc_sharded(be, args, rank, world, use_pg)
            out['extra']['c4_dmc_sharded'] = dict(
                s, note='the --gpus N > 1 headline workload on ONE GPU: the '
                        'strong-scaling reference point')
            curves[key_d] = ref_d = s['walker_steps_per_s']
        if not args.no_cpu and rank == 0:
            spec = box_spec(n)
            out['cpu_baseline'] = cpu_baseline(
                args, spec, n, 0.25 * spec.well_width,
                start=getattr(be, 'seed_confs', {}).get(n))
    else:
        # ---- headline: ONE sharded DMC population, configs[3] ----
        s = bench_dmc_sharded(be, args, rank, world, use_pg)
        out = sharded_line(args, s, world)
        curves[key_d] = out['value']
        if be.has_vmc and not args.no_vmc_extra:
            m = bench_vmc(be, args, rank, world, use_pg, n, W)
            lv = vmc_line(args, m, n, W, world)
            out['extra']['vmc_weak'] = {
                'workload': lv['config']['workload'],
                'value': lv['value'], 'ms_per_step': lv['ms_per_step'],
                'scaling': 'weak',
                'energy_per_particle': m['energy_per_particle'],
                'accept_rate': m['accept_rate'],
            }
            curves[key_v] = lv['value']
            m['vmc'].close()
        if not args.no_scaling_ref:
            # The 1-GPU reference points, in this run, on this box: rank 0
            # alone runs the WHOLE population of configs[3] (and one rank's
            # VMC share) through the same code while the other ranks wait at
            # the barrier below.
            ref_how = ('same run: rank 0 alone on its GPU, the other ranks '
                       'idle at a barrier')
            if rank == 0:
                s1 = bench_dmc_sharded(be, args, 0, 1, False, solo=True)
                ref_d = s1['walker_steps_per_s']
                ref_detail = {k: s1.get(k) for k in (
                    'ms_per_step', 'evolve_kernel_ms', 'mean_walkers',
                    'energy_per_particle', 'hbm_achieved_GBs')}
                if be.has_vmc and not args.no_vmc_extra:
                    m1 = bench_vmc(be, args, 0, 1, False, n, W)
                    ref_v = W * args.steps / m1['dt']
                    m1['vmc'].close()
            dist.barrier()
            be.sync()
    out['extra']['curves'] = curves
    out['extra']['curves_note'] = (
        f'{key_v}: VMC N={n}, {W} chains PER GPU, whole-job walker-steps/s '
        f'(weak scaling); {key_d}: DMC N={args.c4_bosons}, ONE population of '
        f'{args.c4_walkers} walkers over all GPUs (strong scaling); same keys '
        f'at every --gpus, `value` is {key_v if world == 1 else key_d} here')
    out['extra']['strong_scaling'] = scaling_entry(
        key_d, curves[key_d], ref_d, world, ref_how, ref_detail)
    out['extra']['weak_scaling'] = scaling_entry(
        key_v, curves[key_v], ref_v, world, ref_how, None)
    # which engine produced the line (the CPU stand-in of tests/ says so)
    out['backend'] = getattr(be, 'name', 'hip')
    out['data'] = getattr(be, 'data', 'synthetic')
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    os.close(json_fd)
    if use_pg:
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(launch_ranks(args, argv))
    run_rank(args)


if __name__ == '__main__':
    main()
