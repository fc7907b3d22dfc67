#!/usr/bin/env python
"""bench.py -- walker-steps/s of the MI355X walker-propagation engine.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE
JSON line on rank 0.  A "step" is one pass of the hot path over the whole
ensemble: one all-particle Metropolis step (VMC) of every chain, including the
local-energy evaluation.  The workload is BASELINE.json configs[1]: mrbp_qmc
VMC, N = 64 bosons, 2^20 chains per GPU ("mrbp_qmc box").
Chains are independent, so ranks shard them with no data-path collective
(weak scaling: 2^20 chains on every GPU); only the final block sums are
all-reduced.  `extra.dmc` reports the DMC configuration (configs[2], N = 64,
2^18 target walkers) measured in the same run on rank 0's GPU.

Inputs are resident in HBM when the timed region starts.  `roofline` is the
HBM view the metric contract asks for (algorithmic bytes of SURVEY.md 8d over
the measured kernel time); the path is fp64-VALU bound, so `extra.valu`
gives the pair-evaluation rate as well.  `cpu_baseline` times the CPU oracle
(C restatement of the reference algorithm, OpenMP over chains) on a bounded
sample of the same workload, rank 0 only.
"""
import argparse
import json
import os
import sys
import time
from math import pi

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6        # vector fp64, MI355X datasheet


def box_spec(n):
    from phd_qmclib_amd.mrbp_qmc import Spec
    return Spec(lattice_depth=5 * pi ** 2, lattice_ratio=1,
                interaction_strength=2, boson_number=n, supercell_size=n,
                tbf_contact_cutoff=0.25 * n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=64)
    ap.add_argument('--warmup', type=int, default=16)
    ap.add_argument('--block', type=int, default=16,
                    help='Metropolis steps enqueued per block call')
    ap.add_argument('--bosons', type=int, default=64)
    ap.add_argument('--chains', type=int, default=1 << 20,
                    help='VMC chains per GPU')
    ap.add_argument('--dmc-walkers', type=int, default=1 << 18)
    ap.add_argument('--no-dmc', action='store_true')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--dmc-dist', action='store_true',
                    help='also time ONE DMC population sharded over all ranks '
                         '(per-step RCCL all-reduce of (E_t, W_t) + rebalance)')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP engine has no CPU path)')
    torch.cuda.set_device(local_rank)
    use_pg = world > 1 or 'RANK' in os.environ      # launched by torchrun
    if use_pg:
        dist.init_process_group('nccl', device_id=torch.device('cuda',
                                                               local_rank))

    from phd_qmclib_amd.engine import DmcEnsemble, ModelEngine, VmcEnsemble

    n = args.bosons
    W = args.chains
    spec = box_spec(n)
    cfc = spec.cfc_spec
    stream = torch.cuda.current_stream().cuda_stream
    eng = ModelEngine(cfc, device=local_rank, stream=stream)

    def barrier():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------- VMC, configs[1] ----------------
    move_spread = 0.25 * spec.well_width
    rng = np.random.RandomState(1000 + rank)
    pos = spec.supercell_size * rng.random_sample((W, n))
    vmc = VmcEnsemble(eng, W, move_spread, rng_seed=1, chain0=rank * W)
    vmc.set_state(pos)
    del pos

    def run_steps(k):
        # one kernel launch per Metropolis step (the block call enqueues
        # `b` launches without a host synchronisation)
        done = 0
        while done < k:
            b = min(args.block, k - done)
            vmc.run_block(b, sums=False)
            done += b
        return k

    run_steps(args.warmup)
    barrier()
    eng.timer_start()
    t0 = time.perf_counter()
    launches = run_steps(args.steps)
    kernel_ms = eng.timer_stop()          # HIP events on the launch stream
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device='cuda')
    if use_pg:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # spread of the launch time (SURVEY.md 8d asks for median and min): a few
    # more chunks, each bracketed by its own HIP events, outside the timed region
    chunk_ms = []
    for _ in range(6):
        eng.timer_start()
        vmc.run_block(args.block, sums=False)
        chunk_ms.append(eng.timer_stop() / args.block)
    chunk_ms.sort()

    # block estimators of the last block (global reduction over ranks)
    se_ptr, se2_ptr, na_ptr = vmc.block_sums_dev()
    res = vmc.run_block(args.block, sums=True)
    tot = torch.tensor([res['sum_energy'].sum(), float(res['num_accepted'].sum()),
                        float(W * args.block)], dtype=torch.float64,
                       device='cuda')
    if use_pg:
        dist.all_reduce(tot)
    tot = tot.cpu().numpy()

    value = world * W * args.steps / dt
    b_vmc = 16 * n + 32                      # SURVEY.md 8(d), bytes/chain-step
    launch_ms = kernel_ms / launches
    steps_per_launch = args.steps / launches
    achieved = W * steps_per_launch * b_vmc / (launch_ms * 1e-3) / 1e9
    pairs = n * (n - 1) // 2
    pair_rate = W * args.steps * pairs / (kernel_ms * 1e-3)

    # measured HBM traffic per chain-step (rocprofv3 FETCH_SIZE/WRITE_SIZE passes,
    # profiles/r01_traffic.json), scaled to the units of one launch
    traffic = None
    valu_per_step = None
    try:
        with open(os.path.join(ROOT, 'profiles', 'r01_traffic.json')) as fp:
            tj = json.load(fp)
        if n == 64:
            traffic = tj['vmc_step_kernel_bytes_per_chain_step_N64'] * W * \
                steps_per_launch
            valu_per_step = tj.get(
                'vmc_step_kernel_valu_instr_per_chain_step_N64')
    except (OSError, KeyError, ValueError):
        pass

    out = {
        'metric': 'walker-steps/sec',
        'value': value,
        'unit': 'walker-steps/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': {
            'workload': f'mrbp_qmc VMC, N={n} bosons, {W} chains per GPU, '
                        f'move_spread=0.25*well_width, energy on accepted moves',
            'bosons': n, 'chains_per_gpu': W, 'steps_per_launch': 1,
            'parallelism': f'chains sharded over {world} GPU(s), no data-path '
                           f'collective',
        },
        'roofline': {
            'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
            'kernel': 'vmc_step_kernel', 'launch_ms': launch_ms,
            'bytes_per_unit': b_vmc,
        },
        'extra': {
            'valu': {'pair_evals_per_s': pair_rate,
                     # the ceiling that binds: wave64 VALU instructions issued
                     # (SQ_INSTS_VALU / SQ_WAVES of the rocprofv3 PMC pass,
                     # profiles/r01_traffic.json) against 1 instruction per 4
                     # cycles per SIMD, 4 SIMDs x 256 CUs at 2.4 GHz
                     'instr_per_chain_step': valu_per_step,
                     'issue_frac': None if valu_per_step is None else
                     (W * steps_per_launch * valu_per_step /
                      (launch_ms * 1e-3)) / (1024 * 2.4e9 / 4),
                     'note': 'the path is fp64-VALU bound (SURVEY.md 8d); '
                             'unique pairs N(N-1)/2 per chain-step'},
            'launch_ms_min': chunk_ms[0],
            'launch_ms_median': 0.5 * (chunk_ms[2] + chunk_ms[3]),
            'vmc_energy_per_particle': float(tot[0] / tot[2] / n),
            'vmc_accept_rate': float(tot[1] / tot[2]),
        },
    }

    # ---------------- DMC, configs[2] (rank 0's GPU, informational) -------
    if not args.no_dmc and rank == 0:
        target = args.dmc_walkers
        maxw = ((target * 512 // 480) + 255) // 256 * 256
        pos0, _, _ = vmc.get_state()
        d = DmcEnsemble(eng, 6.25e-4, maxw, target, 0.5, rng_seed=1)
        d.set_state(pos0[:target])
        del pos0
        d.run_block(args.warmup, read=False)
        eng.sync()
        eng.timer_start()
        t0 = time.perf_counter()
        d.run_block(args.steps, read=False)
        dmc_ms = eng.timer_stop()
        ddt = time.perf_counter() - t0
        ser = d.read_series(args.steps)
        nws = float(ser.num_walkers.sum())
        b_dmc = 32 * n + 40 + 16
        out['extra']['dmc'] = {
            'workload': f'mrbp_qmc DMC, N={n}, target {target} / max {maxw} '
                        f'walkers, dt=6.25e-4',
            'walker_steps_per_s': nws / ddt,
            'ms_per_step': ddt / args.steps * 1e3,
            'hbm_achieved_GBs': nws * b_dmc / (dmc_ms * 1e-3) / 1e9,
            'hbm_frac': nws * b_dmc / (dmc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            'mean_walkers': nws / args.steps,
            'energy_per_particle': float(ser.energy.sum() / ser.weight.sum() / n),
        }
        d.close()

    # ---------------- DMC, one population over all ranks (opt-in) ---------
    if args.dmc_dist:
        from phd_qmclib_amd.dist import DistributedDmc
        per_rank = args.dmc_walkers
        target = per_rank * world
        cap = ((per_rank * 512 // 480) + 255) // 256 * 256
        pos0, _, _ = vmc.get_state()
        d = DmcEnsemble(eng, 6.25e-4, cap, target, 0.5, rng_seed=1,
                        slot0=rank * cap, external_reduce=True)
        d.set_state(pos0[:per_rank])
        # every rank must start from the same E_ref: the global mean energy
        er = torch.tensor([d.get_state().ref_energy], dtype=torch.float64,
                          device='cuda')
        if use_pg:
            dist.all_reduce(er)
        d.set_state(pos0[:per_rank], ref_energy=float(er.item()) / world)
        del pos0
        dd = DistributedDmc(d, n, torch.device('cuda', local_rank),
                            rebalance_every=32)
        dd.run_block(args.warmup)
        barrier()
        t0 = time.perf_counter()
        ser = dd.run_block(args.steps)
        barrier()
        ddt = time.perf_counter() - t0
        loc = torch.tensor([float(ser.num_walkers.sum()), ddt],
                           dtype=torch.float64, device='cuda')
        tmx = loc[1:].clone()
        if use_pg:
            dist.all_reduce(loc[:1])
            dist.all_reduce(tmx, op=dist.ReduceOp.MAX)
        if rank == 0:
            out['extra']['dmc_dist'] = {
                'workload': f'mrbp_qmc DMC, N={n}, ONE population of target '
                            f'{target} walkers sharded over {world} rank(s), '
                            f'16-byte all-reduce per step, rebalance every 32',
                'walker_steps_per_s': float(loc[0].item()) / float(tmx.item()),
                'ms_per_step': float(tmx.item()) / args.steps * 1e3,
                'walkers_moved_rank0': dd.walkers_moved,
                'energy_per_particle':
                    float(ser.energy.sum() / ser.weight.sum() / n),
            }
        d.close()

    # ---------------- CPU baseline (oracle, rank 0, N = 1 only) -----------
    if not args.no_cpu and rank == 0 and world == 1:
        from oracle import qmc_oracle as orc
        m = orc.model_from_cfc(cfc)
        # the one-GPU box shares its host: use its CPU allotment, not every
        # hardware thread the kernel reports
        cores = max(1, min(orc.max_threads(), len(os.sched_getaffinity(0)), 16))
        rng = np.random.RandomState(7)
        wc, ns = 64 * cores, 4
        cpos = spec.supercell_size * rng.random_sample((wc, n))
        cwf = np.array([orc.wf_abs_log(m, cpos[i]) for i in range(wc)])
        cec = np.zeros(wc)
        t0 = time.perf_counter()
        orc.vmc_ensemble(m, cpos, cwf, cec, move_spread, 1, ns,
                         yield_initial=True, nthreads=cores)
        probe = time.perf_counter() - t0
        # scale the sample to about cpu_seconds of work
        ns2 = max(4, int(ns * args.cpu_seconds / max(probe, 1e-3)))
        t0 = time.perf_counter()
        orc.vmc_ensemble(m, cpos, cwf, cec, move_spread, 1, ns2, step0=ns,
                         nthreads=cores)
        cdt = time.perf_counter() - t0
        out['cpu_baseline'] = {
            'value': wc * ns2 / cdt, 'unit': 'walker-steps/s', 'cores': cores,
            'kind': 'port',
            'sample': f'{wc} chains x {ns2} steps of the same VMC workload '
                      f'(N={n}), oracle/qmc_oracle.c with OpenMP over chains',
        }

    if rank == 0:
        print(json.dumps(out))
    vmc.close()
    eng.close()
    if use_pg:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
