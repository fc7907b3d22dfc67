"""Multi-GPU drivers: one process per GPU, `torch.distributed` (backend "nccl"
= RCCL over xGMI on ROCm; "gloo" in the CPU tests).

What is exchanged, and why only that:

* VMC chains are independent: ranks shard them (Philox stream = global chain
  index) and only all-reduce 4 scalars per block (sum E, sum E^2, accepted,
  steps).
* DMC has exactly one data dependency per time step: the population-control
  feedback E_ref = <E> - kappa ln(W_t / target) / dt needs the GLOBAL E_t and
  W_t (qmc_base/dmc.py:759-771).  Each rank branches and propagates its own
  walkers (`step_local`), leaves (E_t, W_t) in a 2-double device buffer, the
  buffer is all-reduced in place (16 bytes, latency-bound; xGMI bandwidth is
  irrelevant), and `step_finish` applies the feedback with the global sums --
  every rank computes the same E_ref.  All three are enqueued on ONE stream
  (the engine must have been created on the stream torch is using:
  `ModelEngine(..., stream=torch.cuda.current_stream().cuda_stream)`; checked
  here), so the collective is ordered after `step_local` and before
  `step_finish` by the stream itself, the host runs ahead of the device and
  never synchronises inside a block.
* Local populations random-walk apart, so every `rebalance_every` steps the
  ranks all-gather their counts (the one host synchronisation of a
  rebalance), derive the same greedy plan (ranks above the mean send their
  tail walkers to ranks below it) and move whole walker records (pos, drift,
  lane labels, energy, weight and the walker's forward-walking estimator rows)
  with point-to-point send/recv -- single hop on the fully connected xGMI
  mesh.  Packing, transfer, unpacking and the new population size are all
  stream-ordered.
* The S(k) / density estimators are sums over walkers: every rank evaluates
  them on its own walkers (`step_estimators`) and the per-step rows are
  all-reduced once per block.

The reference has no distributed path at all; its global cap
`max_num_walkers` becomes a per-rank cap max_num_walkers / world here.
"""
import ctypes as C
import time
import typing as t

import numpy as np
import torch
import torch.distributed as dist

__all__ = ['DistributedDmc', 'DistributedVmc', 'rebalance_plan']


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def rebalance_plan(counts: t.Sequence[int]) -> t.List[t.Tuple[int, int, int]]:
    """Deterministic transfer list [(src, dst, n), ...] that levels `counts`
    to within one walker: rank r ends with total // G (+1 for r < total % G).
    Every rank derives the same plan from the all-gathered counts."""
    counts = [int(c) for c in counts]
    G, total = len(counts), sum(counts)
    want = [total // G + (1 if r < total % G else 0) for r in range(G)]
    surplus = [[r, counts[r] - want[r]] for r in range(G)
               if counts[r] > want[r]]
    deficit = [[r, want[r] - counts[r]] for r in range(G)
               if counts[r] < want[r]]
    plan = []
    i = j = 0
    while i < len(surplus) and j < len(deficit):
        n = min(surplus[i][1], deficit[j][1])
        plan.append((surplus[i][0], deficit[j][0], n))
        surplus[i][1] -= n
        deficit[j][1] -= n
        if surplus[i][1] == 0:
            i += 1
        if deficit[j][1] == 0:
            j += 1
    return plan


class _RawDeviceArray:
    """A device address as a zero-copy torch tensor source
    (`__cuda_array_interface__`; ROCm builds of torch use the same name)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {
            'shape': (int(count),), 'typestr': '<f8',
            'data': (int(ptr), False), 'version': 2, 'strides': None}


def _wrap_f64(ptr: int, count: int, device: torch.device) -> torch.Tensor:
    """View `count` doubles at address `ptr` as a tensor on `device`."""
    if device.type == 'cuda':
        return torch.as_tensor(_RawDeviceArray(ptr, count), device=device)
    arr = np.ctypeslib.as_array(
        C.cast(ptr, C.POINTER(C.c_double)), shape=(int(count),))
    return torch.from_numpy(arr)


def _wrap_i64(ptr: int, count: int, device: torch.device) -> torch.Tensor:
    """View `count` 64-bit integers at device address `ptr`."""
    src = _RawDeviceArray(ptr, count)
    src.__cuda_array_interface__['typestr'] = '<i8'
    return torch.as_tensor(src, device=device)


class DistributedVmc:
    """Chains sharded over ranks; per-block global sums by all-reduce."""

    def __init__(self, sampling_factory, chains_per_rank: int, device=None):
        """sampling_factory(first_chain) -> an object with `.ensemble`
        (`run_block`, `block_sums_dev`) e.g. `vmc.EnsembleSampling`."""
        self.rank, self.world = _world()
        self.chains_per_rank = int(chains_per_rank)
        self.sampling = sampling_factory(self.rank * self.chains_per_rank)
        self.device = device

    def run_block(self, num_steps: int) -> t.Dict[str, float]:
        """-> global block statistics (identical on every rank).  The
        per-chain sums of a GPU ensemble are reduced where they are (24 bytes
        per chain stay in HBM; four doubles cross to the other ranks and to the
        host); the CPU stand-in of the tests hands back host arrays."""
        ens = self.sampling.ensemble
        dev = None if self.device is None else torch.device(self.device)
        n_samples = float(self.chains_per_rank * num_steps)
        if dev is not None and dev.type == 'cuda' and \
                hasattr(ens, 'block_sums_dev'):
            ens.run_block(int(num_steps), sums=False)
            # the engine may launch on a stream of its own: its block is
            # complete before torch reads the sums
            ens.engine.sync()
            W = self.chains_per_rank
            p_e, p_e2, p_acc = ens.block_sums_dev()
            loc = torch.stack([
                _wrap_f64(p_e, W, dev).sum(), _wrap_f64(p_e2, W, dev).sum(),
                _wrap_i64(p_acc, W, dev).sum().to(torch.float64),
                torch.tensor(n_samples, dtype=torch.float64, device=dev)])
        else:
            out = ens.run_block(int(num_steps))
            loc = torch.tensor([out['sum_energy'].sum(),
                                out['sum_energy2'].sum(),
                                float(out['num_accepted'].sum()), n_samples],
                               dtype=torch.float64, device=self.device)
        if self.world > 1:
            dist.all_reduce(loc)
        se, se2, na, n = loc.tolist()
        return dict(energy_mean=se / n, energy2_mean=se2 / n,
                    accept_rate=na / n, num_samples=n)


class DistributedDmc:
    """One DMC population sharded over the ranks of the default process group.

    `ensemble` is this rank's population handle (`engine.DmcEnsemble` created
    with `external_reduce=True`, the GLOBAL target and its local cap); the CPU
    tests pass an oracle-backed stand-in with the same methods.  `device` is
    the torch device of the communication buffers ('cuda:<local_rank>' for
    RCCL, 'cpu' for gloo).
    """

    def __init__(self, ensemble, num_particles: int, device,
                 rebalance_every: int = 32, imbalance_tol: float = 0.02,
                 force_collectives: bool = False, solo: bool = False):
        """`force_collectives` issues the per-step all-reduce even in a group
        of one rank (tests: the RCCL call and its stream ordering run on a
        single GPU; a one-rank all-reduce is the identity).  `solo`: the
        population lives on this rank alone whatever the process group's size
        (bench.py: the one-GPU strong-scaling reference timed by rank 0 inside
        a multi-rank run); no collective is issued."""
        self.ens = ensemble
        self.n = int(num_particles)
        self.device = torch.device(device)
        self.rank, self.world = (0, 1) if solo else _world()
        self.rebalance_every = int(rebalance_every)
        self.imbalance_tol = float(imbalance_tol)
        self._collect = not solo and (self.world > 1 or (
            force_collectives and dist.is_available() and dist.is_initialized()))
        self._check_stream()
        self.sums = torch.zeros(2, dtype=torch.float64, device=self.device)
        self.steps_done = 0
        self.walkers_moved = 0
        self.rebalances = 0
        self._inflight = []      # transfer buffers of the last rebalance
        self._phase = None       # per-phase timings (enable_phase_timing)

    def _check_stream(self):
        """The collectives are ordered with the engine's kernels only when
        both use the same stream."""
        if self.device.type != 'cuda':
            return
        engine = getattr(self.ens, 'engine', None)
        if engine is None:
            return
        mine = int(engine.stream_handle)
        theirs = int(torch.cuda.current_stream(self.device).cuda_stream)
        if mine != theirs:
            raise RuntimeError(
                f'DistributedDmc: the engine launches on stream {mine:#x} but '
                f'torch.cuda.current_stream() is {theirs:#x}; create the '
                f'engine with stream=torch.cuda.current_stream().cuda_stream '
                f'(and keep that stream current) so that RCCL collectives are '
                f'ordered with the walker kernels')

    # the handle gets raw addresses; tensors stay alive on self
    def _ptr(self, tensor):
        return tensor.data_ptr()

    # ---- per-phase timings (bench.py `extra.phases`) ----------------------
    def enable_phase_timing(self, max_steps: int = 4096):
        """From now on record, per time step, the duration of the all-reduce
        on the stream (an event pair around it; wall clock with gloo) and the
        host time spent enqueueing the step, and per rebalance its wall time.
        Costs two event records per step; read with `phase_report`."""
        ph = dict(steps=0, host_enqueue_s=0.0, allreduce_wall_s=0.0,
                  rebalance_s=0.0, rebalance_calls=0, events=[], free=[])
        if self.device.type == 'cuda' and self._collect:
            ph['free'] = [torch.cuda.Event(enable_timing=True)
                          for _ in range(2 * int(max_steps))]
        self._phase = ph

    def phase_report(self) -> t.Optional[t.Dict[str, float]]:
        """Synchronises.  -> dict(steps, allreduce_us_per_step,
        allreduce_us_max, host_enqueue_us_per_step, rebalance_ms_total,
        rebalance_calls) since `enable_phase_timing`, or None."""
        ph = self._phase
        if ph is None:
            return None
        steps = max(ph['steps'], 1)
        ar = []
        if ph['events']:
            torch.cuda.synchronize(self.device)
            ar = [a.elapsed_time(b) * 1e3 for a, b in ph['events']]   # us
        # steps beyond the event pool were timed on the host clock: the mean
        # covers every step (events where there were events, wall clock for the
        # rest), and the line says how many steps each clock covered
        wall_steps = ph['steps'] - len(ar)
        mean_ar = (sum(ar) + ph['allreduce_wall_s'] * 1e6) / steps
        how = 'hip events on the stream' if ar else 'host wall clock'
        if ar and wall_steps > 0:
            how = (f'hip events on the stream ({len(ar)} steps) + host wall '
                   f'clock ({wall_steps} steps: event pool exhausted)')
        return dict(steps=ph['steps'], allreduce_us_per_step=mean_ar,
                    allreduce_us_max=max(ar) if ar else None,
                    allreduce_event_timed_steps=len(ar),
                    allreduce_timed_with=how,
                    host_enqueue_us_per_step=ph['host_enqueue_s'] * 1e6 / steps,
                    rebalance_ms_total=ph['rebalance_s'] * 1e3,
                    rebalance_calls=ph['rebalance_calls'])

    def step(self):
        """One global time step, fully enqueued (no host synchronisation)."""
        ph = self._phase
        t0 = time.perf_counter() if ph is not None else 0.0
        self.ens.step_local(self._ptr(self.sums))
        if self._collect:
            if ph is not None and len(ph['free']) >= 2:
                e0, e1 = ph['free'].pop(), ph['free'].pop()
                e0.record()
                dist.all_reduce(self.sums)      # 16 bytes, in place
                e1.record()
                ph['events'].append((e0, e1))
            elif ph is not None:
                t1 = time.perf_counter()
                dist.all_reduce(self.sums)
                ph['allreduce_wall_s'] += time.perf_counter() - t1
            else:
                dist.all_reduce(self.sums)      # 16 bytes, in place
        self.ens.step_finish(self._ptr(self.sums))
        self.steps_done += 1
        if ph is not None:
            ph['steps'] += 1
            ph['host_enqueue_s'] += time.perf_counter() - t0

    def run_block(self, num_steps: int, estimators: bool = False):
        """`num_steps` time steps with periodic population rebalance;
        -> the per-step series (global E_t, W_t; local walker counts), and
        with `estimators` also (iter_ssf[num_steps, M, 3] or None,
        iter_density[num_steps, B] or None) summed over all ranks."""
        num_steps = int(num_steps)
        if estimators:
            self.ens.est_begin_block(num_steps)
        for t_idx in range(num_steps):
            if (self.world > 1 and self.rebalance_every > 0 and
                    self.steps_done % self.rebalance_every == 0 and
                    self.steps_done > 0):
                self.rebalance()
            self.step()
            if estimators:
                self.ens.step_estimators(t_idx)
        if not estimators:
            return self.ens.read_series(num_steps)
        ssf, dens = self._reduce_estimators(num_steps)
        return self.ens.read_series(num_steps), ssf, dens

    def _reduce_estimators(self, num_steps: int):
        M = int(getattr(self.ens, 'num_modes', 0))
        B = int(getattr(self.ens, 'num_bins', 0))
        p_ssf, p_dens = self.ens.est_iter_dev()
        out = []
        for ptr, cnt, shape in ((p_ssf, num_steps * M * 3, (num_steps, M, 3)),
                                (p_dens, num_steps * B, (num_steps, B))):
            if not ptr or not cnt:
                out.append(None)
                continue
            view = _wrap_f64(ptr, cnt, self.device)
            if self._collect:
                dist.all_reduce(view)           # in place, engine's stream
            out.append(view.cpu().numpy().reshape(shape).copy())
        return out[0], out[1]

    def local_count(self) -> int:
        return int(self.ens.num_walkers())

    def global_counts(self) -> t.List[int]:
        """Synchronising: reads the device walker count."""
        if self.world == 1:
            return [self.local_count()]
        mine = torch.tensor([self.local_count()], dtype=torch.int64,
                            device=self.device)
        allc = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(allc, mine)
        return [int(c.item()) for c in allc]

    def record_size(self) -> int:
        f = getattr(self.ens, 'walker_record_size', None)
        return int(f()) if f is not None else 3 * self.n + 2

    def rebalance(self, force: bool = False) -> int:
        """Level the local populations; -> number of walkers this rank sent
        or received.  One host synchronisation (the walker counts); packing,
        transfers, unpacking and the new population size are stream-ordered."""
        t_start = time.perf_counter()
        counts = self.global_counts()
        self.last_counts = counts          # before any transfer
        mean = sum(counts) / len(counts)
        if not force and mean > 0 and \
                (max(counts) - min(counts)) <= self.imbalance_tol * mean:
            if self._phase is not None:     # (its all-gather was paid for)
                self._phase['rebalance_s'] += time.perf_counter() - t_start
                self._phase['rebalance_calls'] += 1
            return 0
        plan = rebalance_plan(counts)
        rec = self.record_size()
        moved = 0
        nw = counts[self.rank]
        ops, recvs, keep = [], [], []
        for src, dst, cnt in plan:
            if src == self.rank:
                buf = torch.empty(cnt * rec, dtype=torch.float64,
                                  device=self.device)
                # send the tail of the local population
                self.ens.export_walkers(nw - cnt, cnt, self._ptr(buf))
                nw -= cnt
                ops.append(dist.P2POp(dist.isend, buf, dst))
                keep.append(buf)
                moved += cnt
            elif dst == self.rank:
                buf = torch.empty(cnt * rec, dtype=torch.float64,
                                  device=self.device)
                ops.append(dist.P2POp(dist.irecv, buf, src))
                recvs.append((buf, cnt))
                keep.append(buf)
                moved += cnt
        if ops:
            # RCCL: the transfers wait for the packing kernels through the
            # current stream, and wait() makes that stream wait for them
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if nw != counts[self.rank]:
            self.ens.set_num_walkers(nw)
        for buf, cnt in recvs:
            self.ens.import_walkers_at(nw, cnt, self._ptr(buf))
            nw += cnt
        # the unpack kernels read the buffers asynchronously: keep them until
        # the next rebalance (by then the stream has long passed them)
        self._inflight = keep
        self.walkers_moved += moved
        self.rebalances += 1 if moved else 0
        if self._phase is not None:
            self._phase['rebalance_s'] += time.perf_counter() - t_start
            self._phase['rebalance_calls'] += 1
        return moved
