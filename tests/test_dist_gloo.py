"""world_size-2 tests of the multi-GPU host logic on CPU (gloo backend): the
E_ref feedback sees global sums and is identical on every rank; the population
rebalance conserves walkers and levels the counts."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from phd_qmclib_amd.dist import rebalance_plan
from .conftest import ROOT


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def test_rebalance_plan_levels_and_conserves():
    rng = np.random.RandomState(0)
    for _ in range(200):
        G = rng.randint(1, 9)
        counts = list(rng.randint(0, 1000, size=G))
        plan = rebalance_plan(counts)
        new = list(counts)
        for src, dst, n in plan:
            assert n > 0 and src != dst
            new[src] -= n
            new[dst] += n
        assert sum(new) == sum(counts)
        assert max(new) - min(new) <= 1
        # a rank never both sends and receives
        assert not ({s for s, _, _ in plan} & {d for _, d, _ in plan})


@pytest.mark.timeout(300)
def test_two_rank_dmc_gloo(tmp_path, oracle):
    port = free_port()
    procs = [subprocess.Popen([sys.executable,
                               os.path.join(ROOT, 'tests', '_dist_worker.py'),
                               str(r), '2', port, str(tmp_path)])
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    r0, r1 = [json.load(open(tmp_path / f'rank{r}.json')) for r in range(2)]
    # forced rebalance: 36 + 12 -> 24 + 24, 12 walkers moved, none lost
    assert r0['counts_before'] == [36, 12] and r0['counts_after'] == [24, 24]
    assert r0['moved'] == 12 and r1['moved'] == 12
    before = sorted(r0['fp_before'] + r1['fp_before'])
    after = sorted(r0['fp_after'] + r1['fp_after'])
    assert before == after
    s0, s1 = np.array(r0['series']), np.array(r1['series'])
    # global E_t, W_t, E_ref, accum identical on both ranks every step
    assert np.array_equal(s0[:, [0, 1, 3, 4]], s1[:, [0, 1, 3, 4]])
    # W_t is the sum of the local populations
    assert np.array_equal(s0[:, 2] + s1[:, 2], s0[:, 1])
    # energy per particle is physical (box8: E/N ~ 15-16)
    e_per = s0[:, 0] / s0[:, 1] / 8
    assert np.all((e_per > 10) & (e_per < 25))
    assert sum(r0['counts_end']) == int(s0[-1, 1])
    # VMC: both ranks hold the same global block statistics, equal to those
    # of one process running all twelve chains (Philox stream = chain index)
    assert r0['vmc'] == r1['vmc']
    for blk, single in zip(r0['vmc'], r0['vmc_single']):
        assert blk['num_samples'] == 12 * 16
        assert blk['energy_mean'] == pytest.approx(single['energy_mean'],
                                                   rel=1e-13)
        assert blk['accept_rate'] == single['accept_rate']
