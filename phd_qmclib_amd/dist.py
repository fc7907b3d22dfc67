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
  every rank computes the same E_ref.  All three are enqueued on one stream,
  so the host runs ahead of the device and never synchronises inside a block.
* Local populations random-walk apart, so every `rebalance_every` steps the
  ranks all-gather their counts, derive the same greedy plan (ranks above the
  mean send their tail walkers to ranks below it) and move whole walker
  records (pos, drift, lane labels, energy, weight: 3N+2 doubles) with point-to-point
  send/recv -- single hop on the fully connected xGMI mesh.

The reference has no distributed path at all; its global cap
`max_num_walkers` becomes a per-rank cap max_num_walkers / world here.
"""
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
        """-> global block statistics (identical on every rank)."""
        out = self.sampling.ensemble.run_block(int(num_steps))
        loc = torch.tensor([out['sum_energy'].sum(),
                            out['sum_energy2'].sum(),
                            float(out['num_accepted'].sum()),
                            float(self.chains_per_rank * num_steps)],
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
                 rebalance_every: int = 32, imbalance_tol: float = 0.02):
        self.ens = ensemble
        self.n = int(num_particles)
        self.device = torch.device(device)
        self.rank, self.world = _world()
        self.rebalance_every = int(rebalance_every)
        self.imbalance_tol = float(imbalance_tol)
        self.sums = torch.zeros(2, dtype=torch.float64, device=self.device)
        self.steps_done = 0
        self.walkers_moved = 0

    # the handle gets raw addresses; tensors stay alive on self
    def _ptr(self, tensor):
        return tensor.data_ptr()

    def step(self):
        """One global time step, fully enqueued (no host synchronisation)."""
        self.ens.step_local(self._ptr(self.sums))
        if self.world > 1:
            dist.all_reduce(self.sums)          # 16 bytes, in place
        self.ens.step_finish(self._ptr(self.sums))
        self.steps_done += 1

    def run_block(self, num_steps: int):
        """`num_steps` time steps with periodic population rebalance;
        -> the per-step series (global E_t, W_t; local walker counts)."""
        for _ in range(int(num_steps)):
            if (self.world > 1 and self.rebalance_every > 0 and
                    self.steps_done % self.rebalance_every == 0 and
                    self.steps_done > 0):
                self.rebalance()
            self.step()
        return self.ens.read_series(int(num_steps))

    def local_count(self) -> int:
        return int(self.ens.num_walkers())

    def global_counts(self) -> t.List[int]:
        if self.world == 1:
            return [self.local_count()]
        mine = torch.tensor([self.local_count()], dtype=torch.int64,
                            device=self.device)
        allc = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(allc, mine)
        return [int(c.item()) for c in allc]

    def rebalance(self, force: bool = False) -> int:
        """Level the local populations; -> number of walkers this rank sent
        or received.  Synchronising (reads the device walker count)."""
        counts = self.global_counts()
        mean = sum(counts) / len(counts)
        if not force and mean > 0 and \
                (max(counts) - min(counts)) <= self.imbalance_tol * mean:
            return 0
        plan = rebalance_plan(counts)
        rec = 3 * self.n + 2
        moved = 0
        nw = counts[self.rank]
        ops, bufs = [], []
        for src, dst, cnt in plan:
            if src == self.rank:
                buf = torch.empty(cnt * rec, dtype=torch.float64,
                                  device=self.device)
                # send the tail of the local population
                self.ens.export_walkers(nw - cnt, cnt, self._ptr(buf))
                nw -= cnt
                ops.append(dist.P2POp(dist.isend, buf, dst))
                bufs.append((None, buf, cnt))
                moved += cnt
            elif dst == self.rank:
                buf = torch.empty(cnt * rec, dtype=torch.float64,
                                  device=self.device)
                ops.append(dist.P2POp(dist.irecv, buf, src))
                bufs.append(('recv', buf, cnt))
                moved += cnt
        if self.device.type == 'cuda':
            torch.cuda.current_stream().synchronize()   # packed before send
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if nw != counts[self.rank]:
            self.ens.truncate(nw)
        for kind, buf, cnt in bufs:
            if kind == 'recv':
                self.ens.import_walkers(cnt, self._ptr(buf))
        self.walkers_moved += moved
        return moved
