"""bench.py spawns its own ranks: `python bench.py --gpus 2` (no RANK in the
environment) must run TWO ranks that join one process group and print ONE JSON
line whose n_gpus is the world size the process group reports.  On CPU the
GPU backend is replaced by the gloo / oracle stand-in of tests/_bench_standin
(host logic only; the GPU numbers come from the driver's runs)."""
import json
import os
import subprocess
import sys

import pytest

from .conftest import ROOT


def run_bench(args, env_extra, timeout=240):
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR',
                        'MASTER_PORT')}
    env.update(env_extra)
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] +
                          args, env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.timeout(300)
def test_bench_spawns_two_ranks(oracle):
    r = run_bench(['--gpus', '2', '--steps', '6', '--warmup', '2',
                   '--c4-bosons', '8', '--c4-walkers', '48',
                   '--rebalance-every', '2', '--start-skew', '0.25',
                   '--no-checks'],
                  {'QMC_BENCH_BACKEND': 'tests._bench_standin'})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2                       # two ranks joined
    assert out['scaling'] == 'strong'
    assert out['metric'] == 'walker-steps/sec' and out['value'] > 0
    assert out['steps'] == 6 and out['warmup'] == 2
    assert out['config']['global_target_walkers'] == 48
    assert out['config']['walkers_per_gpu'] == 24
    # W_t of every step is the sum over BOTH ranks: about the global target
    assert 30 < out['extra']['mean_walkers'] < 70
    assert 'cpu_baseline' not in out                # rank 0 at N = 1 only
    # the line says which engine produced it
    assert out['backend'].startswith('cpu-standin')
    assert out['data'] == 'cpu-standin'
    # the ranks started 30 / 18 and the forced rebalances really moved walkers
    # (warm-up: 6 sent + 6 received), conserving the population
    ex = out['extra']
    wu = ex['rebalance_checks']['warmup']
    assert sum(wu['counts_before']) == sum(wu['counts_after'])
    assert max(wu['counts_after']) - min(wu['counts_after']) <= 1
    assert max(wu['counts_before']) - min(wu['counts_before']) >= 6
    assert ex['walkers_moved_warmup_all_ranks'] >= 12
    tm = ex['rebalance_checks']['timed']
    assert sum(tm['counts_before']) == sum(tm['counts_after'])
    assert ex['walkers_moved_all_ranks'] > 0
    assert ex['ref_energy_identical_on_all_ranks'] is True
    ph = ex['phases']
    assert ph['steps'] == 6 and ph['rebalance_calls'] >= 1
    assert ph['allreduce_us_per_step_max_over_ranks'] > 0
    assert ph['host_enqueue_us_per_step_max_over_ranks'] > 0
    assert ph['rebalance_ms_total_max_over_ranks'] > 0
    # both scaling curves under the same keys at every --gpus, and the 1-GPU
    # strong-scaling reference measured in this run (rank 0 alone, the other
    # rank at a barrier): a reader can draw the curve from the lines alone
    assert set(ex['curves']) == {'vmc_n64_weak', 'dmc_n8_strong'}
    assert ex['curves']['dmc_n8_strong'] == out['value']
    assert ex['curves']['vmc_n64_weak'] is None     # (no VMC in the stand-in)
    ss = ex['strong_scaling']
    assert ss['curve'] == 'dmc_n8_strong' and ss['n_gpus'] == 2
    assert ss['ref_1gpu'] > 0 and ss['value'] == out['value']
    assert ss['speedup'] == pytest.approx(out['value'] / ss['ref_1gpu'])
    assert ss['efficiency'] == pytest.approx(ss['speedup'] / 2)
    assert 'same run' in ss['ref_measured']
    assert 30 < ss['ref_detail']['mean_walkers'] < 70   # the WHOLE population
    assert ex['weak_scaling'] is None


def test_bench_refuses_world_size_mismatch():
    # a single process told --gpus 2 by a launcher that started one rank
    r = run_bench(['--gpus', '2', '--steps', '2', '--warmup', '1'],
                  {'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0',
                   'QMC_BENCH_BACKEND': 'tests._bench_standin'})
    assert r.returncode != 0
    assert 'WORLD_SIZE' in r.stderr


@pytest.mark.timeout(120)
def test_launcher_stops_the_other_ranks_when_one_dies(oracle):
    """Rank 1 exits before the rendezvous: rank 0 would wait for it until the
    process group's timeout; the launcher ends it and reports the failure."""
    import time
    t0 = time.time()
    r = run_bench(['--gpus', '2', '--steps', '2', '--warmup', '1',
                   '--c4-bosons', '8', '--c4-walkers', '48', '--no-checks'],
                  {'QMC_BENCH_BACKEND': 'tests._bench_standin',
                   'QMC_STANDIN_FAIL_RANK': '1'}, timeout=100)
    assert r.returncode == 7
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert 'a rank exited with status 7' in r.stderr
    assert time.time() - t0 < 90
