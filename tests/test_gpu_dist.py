"""GPU tests of the multi-GPU building blocks on ONE device (world size 1 and
two in-process shards): stream sharing with torch, the split-step estimators,
and the population rebalance (walker records carry the forward-walking rows,
spare normals are not inherited, everything is stream-ordered)."""
from math import pi

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def box(n=16, **kw):
    from phd_qmclib_amd import mrbp_qmc
    args = dict(lattice_depth=5 * pi ** 2, lattice_ratio=1,
                interaction_strength=2, boson_number=n, supercell_size=n,
                tbf_contact_cutoff=0.25 * n)
    args.update(kw)
    return mrbp_qmc.Spec(**args)


def test_engine_launches_on_the_callers_stream():
    """`stream=0` used to mean "create my own stream", so an
    engine handed torch's default stream silently ran beside it."""
    import torch
    from phd_qmclib_amd.dist import DistributedDmc
    from phd_qmclib_amd.engine import DmcEnsemble, ModelEngine
    cfc = box(16).cfc_spec
    cur = torch.cuda.current_stream().cuda_stream      # 0: legacy default
    e0 = ModelEngine(cfc, stream=cur)
    assert e0.stream_handle == cur and not e0.owns_stream
    own = ModelEngine(cfc)
    assert own.owns_stream and own.stream_handle != 0
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        e1 = ModelEngine(cfc, stream=side.cuda_stream)
        assert e1.stream_handle == side.cuda_stream != 0
        d = DmcEnsemble(e1, 1e-3, 128, 64, 0.5, rng_seed=1,
                        external_reduce=True)
        DistributedDmc(d, 16, 'cuda')                  # same stream: accepted
        d2 = DmcEnsemble(own, 1e-3, 128, 64, 0.5, rng_seed=1,
                         external_reduce=True)
        with pytest.raises(RuntimeError, match='stream'):
            DistributedDmc(d2, 16, 'cuda')             # engine's own stream
        d.close()
        d2.close()
    for e in (e0, own, e1):
        e.close()


def _ensembles(n, nw, maxw, seed, est, **kw):
    import torch
    from phd_qmclib_amd.engine import DmcEnsemble, ModelEngine
    eng = ModelEngine(box(n).cfc_spec,
                      stream=torch.cuda.current_stream().cuda_stream)
    pos = n * np.random.RandomState(seed).random_sample((nw, n))
    d = DmcEnsemble(eng, 1e-3, maxw, nw, 0.5, rng_seed=seed, **kw)
    if est:
        d.set_estimators(**est)
    d.set_state(pos)
    return eng, d


EST = dict(num_modes=12, ssf_pure=True, ssf_pfw=24, num_bins=16,
           dens_pure=True, dens_pfw=24)


def test_split_step_estimators_equal_run_block_est():
    from phd_qmclib_amd.dist import DistributedDmc
    eng_a, a = _ensembles(16, 200, 256, 3, EST)
    eng_b, b = _ensembles(16, 200, 256, 3, EST, external_reduce=True)
    for blk in range(2):                 # per-block resets included
        sa, ssf_a, dens_a = a.run_block_est(20)
        dd = DistributedDmc(b, 16, 'cuda') if blk == 0 else dd
        sb, ssf_b, dens_b = dd.run_block(20, estimators=True)
        assert np.array_equal(sa.energy, sb.energy)
        assert np.array_equal(sa.num_walkers, sb.num_walkers)
        assert np.array_equal(ssf_a, ssf_b)
        assert np.array_equal(dens_a[..., 0], dens_b)
    assert np.abs(ssf_a).max() > 0
    for h in (a, b, eng_a, eng_b):
        h.close()


@pytest.mark.parametrize('when', [7, 8])        # odd and even step counts
def test_rebalance_round_trip_is_exact(when):
    """Export the tail walkers, drop them, import them back in the middle of
    a block with PURE (forward-walking) estimators: every later number must be
    bit-identical to the undisturbed run -- i.e. the walker record carries
    everything a walker owns (positions, drift, lane labels, energy, weight,
    its S(k) and density rows) and an imported slot does not consume a stale
    spare normal."""
    import torch
    from phd_qmclib_amd.dist import DistributedDmc
    res = []
    for disturb in (False, True):
        eng, d = _ensembles(24, 150, 256, 11, EST, external_reduce=True)
        dd = DistributedDmc(d, 24, 'cuda', rebalance_every=0)
        d.est_begin_block(20)
        for t in range(20):
            if disturb and t == when:
                nw = d.num_walkers()
                k = 37
                rec = d.walker_record_size()
                assert rec == 3 * 24 + 2 + 3 * 12 + 16
                buf = torch.zeros(k * rec, dtype=torch.float64, device='cuda')
                d.export_walkers(nw - k, k, buf.data_ptr())
                d.set_num_walkers(nw - k)
                d.import_walkers_at(nw - k, k, buf.data_ptr())
            dd.step()
            d.step_estimators(t)
        ssf, dens = dd._reduce_estimators(20)
        ser = d.read_series(20)
        res.append((ser, ssf, dens))
        d.close()
        eng.close()
    (s0, ssf0, den0), (s1, ssf1, den1) = res
    assert np.array_equal(s0.num_walkers, s1.num_walkers)
    assert np.array_equal(s0.energy, s1.energy)
    assert np.array_equal(s0.ref_energy, s1.ref_energy)
    assert np.array_equal(ssf0, ssf1)
    assert np.array_equal(den0, den1)
    assert np.abs(ssf0[-1]).max() > 0 and den0[-1].sum() > 0


def _fingerprint(h, count):
    """Energies of the first `count` walkers of the CURRENT population."""
    import torch
    rec = h.walker_record_size()
    buf = torch.zeros(count * rec, dtype=torch.float64, device='cuda')
    h.export_walkers(0, count, buf.data_ptr())
    h.engine.sync()
    return buf.cpu().numpy().reshape(count, rec)[:, 3 * h.num_particles]


def test_two_shards_one_population():
    """Two handles on one GPU driven as two ranks of one population: the
    E_ref feedback sees the global sums, a transfer conserves the walkers and
    the mixed S(k) of the union is the sum of the shards' (linearity)."""
    import torch
    eng, a = _ensembles(16, 90, 160, 21, dict(num_modes=8),
                        external_reduce=True, slot0=0)
    pos_b = 16 * np.random.RandomState(22).random_sample((30, 16))
    from phd_qmclib_amd.engine import DmcEnsemble
    b = DmcEnsemble(eng, 1e-3, 160, 90, 0.5, rng_seed=21, slot0=160,
                    external_reduce=True)
    b.set_estimators(num_modes=8)
    b.set_state(pos_b)
    # one target (120) and one E_ref for both shards
    ea, eb = a.get_scalars()[2], b.get_scalars()[2]
    ref = (90 * ea + 30 * eb) / 120
    pos_a = 16 * np.random.RandomState(21).random_sample((90, 16))
    for h, p in ((a, pos_a), (b, pos_b)):
        h.close()
    a = DmcEnsemble(eng, 1e-3, 160, 120, 0.5, rng_seed=21, slot0=0,
                    external_reduce=True)
    b = DmcEnsemble(eng, 1e-3, 160, 120, 0.5, rng_seed=21, slot0=160,
                    external_reduce=True)
    for h, p in ((a, pos_a), (b, pos_b)):
        h.set_estimators(num_modes=8)
        h.set_state(p, ref_energy=ref)
    pa = torch.zeros(2, dtype=torch.float64, device='cuda')
    pb = torch.zeros(2, dtype=torch.float64, device='cuda')
    tot = torch.zeros(2, dtype=torch.float64, device='cuda')
    nsteps = 12
    for h in (a, b):
        h.est_begin_block(nsteps)
    for t in range(nsteps):
        if t == 5:
            # level 90-ish / 30-ish: move 30 tail walkers from a to b
            na, nb = a.num_walkers(), b.num_walkers()
            fp = sorted(np.concatenate([_fingerprint(a, na),
                                        _fingerprint(b, nb)]))
            k = 30
            buf = torch.zeros(k * a.walker_record_size(), dtype=torch.float64,
                              device='cuda')
            a.export_walkers(na - k, k, buf.data_ptr())
            a.set_num_walkers(na - k)
            b.import_walkers_at(nb, k, buf.data_ptr())
            assert a.num_walkers() == na - k and b.num_walkers() == nb + k
            fp2 = sorted(np.concatenate([_fingerprint(a, na - k),
                                         _fingerprint(b, nb + k)]))
            assert fp == fp2
        a.step_local(pa.data_ptr())
        b.step_local(pb.data_ptr())
        torch.add(pa, pb, out=tot)
        a.step_finish(tot.data_ptr())
        b.step_finish(tot.data_ptr())
        a.step_estimators(t)
        b.step_estimators(t)
    sa, sb = a.read_series(nsteps), b.read_series(nsteps)
    # both shards hold the same global series; W_t = sum of the local counts
    assert np.array_equal(sa.energy, sb.energy)
    assert np.array_equal(sa.ref_energy, sb.ref_energy)
    assert np.array_equal(sa.weight,
                          (sa.num_walkers + sb.num_walkers).astype(float))
    # linearity of the mixed S(k): shards' rows add up to the union's
    from phd_qmclib_amd.dist import _wrap_f64
    dev = torch.device('cuda')
    rows = [_wrap_f64(h.est_iter_dev()[0], nsteps * 8 * 3, dev).cpu().numpy()
            .reshape(nsteps, 8, 3) for h in (a, b)]
    sta, stb = a.get_state(), b.get_state()
    confs = np.concatenate([sta.confs[:sta.num_walkers, 0],
                            stb.confs[:stb.num_walkers, 0]])
    k = 2 * np.pi * np.arange(8) / 16
    ph = confs[:, None, :] * k[None, :, None]
    re, im = np.cos(ph).sum(-1), np.sin(ph).sum(-1)
    want = np.stack([re * re + im * im, re, im], axis=-1).sum(0)
    got = rows[0][-1] + rows[1][-1]
    assert np.allclose(got, want, rtol=1e-9, atol=1e-7)
    for h in (a, b, eng):
        h.close()


def test_rccl_all_reduce_is_ordered_with_the_engine_stream():
    """One-rank RCCL group on this GPU: the 16-byte all-reduce really runs
    between step_local and step_finish of every step (force_collectives), on a
    side stream the engine shares with torch.  A one-rank all-reduce is the
    identity, so the series must equal the plain block run bit for bit; if the
    collective were not ordered with the kernels (the engine used to
    launch on a private stream) E_ref would be computed from stale sums."""
    import socket
    import torch
    import torch.distributed as dist
    from phd_qmclib_amd.dist import DistributedDmc
    from phd_qmclib_amd.engine import DmcEnsemble, ModelEngine
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}',
                            rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    try:
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            eng = ModelEngine(box(64).cfc_spec, stream=side.cuda_stream)
            pos = 64 * np.random.RandomState(2).random_sample((4000, 64))
            a = DmcEnsemble(eng, 1e-3, 4608, 4000, 0.5, rng_seed=4)
            b = DmcEnsemble(eng, 1e-3, 4608, 4000, 0.5, rng_seed=4,
                            external_reduce=True)
            b.set_estimators(num_modes=6)
            a.set_state(pos)
            b.set_state(pos)
            sa = a.run_block(40)
            dd = DistributedDmc(b, 64, 'cuda', rebalance_every=8,
                                force_collectives=True)
            assert dd._collect
            sb, ssf, _ = dd.run_block(40, estimators=True)
            assert np.array_equal(sa.num_walkers, sb.num_walkers)
            assert np.array_equal(sa.energy, sb.energy)
            assert np.array_equal(sa.ref_energy, sb.ref_energy)
            assert ssf.shape == (40, 6, 3) and ssf[-1, 1, 0] > 0
            assert dd.global_counts() == [int(sb.num_walkers[-1])]
            a.close(); b.close(); eng.close()
    finally:
        dist.destroy_process_group()


def test_bench_two_ranks_on_one_gpu_real_engine():
    """`bench.py --gpus 2` with the HIP engine on both ranks (sharing this GPU)
    and gloo as transport (tests/_bench_gpu_gloo.py): the multi-rank logic of the
    headline run -- unequal start, split step with the global (E_t, W_t), forced
    rebalances moving real walker records between two processes, conservation,
    bit-identical E_ref on both ranks, phases -- on the product kernels.  What
    it cannot cover is RCCL itself."""
    import json
    import os
    import subprocess
    import sys
    from .conftest import ROOT
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR',
                        'MASTER_PORT')}
    env['QMC_BENCH_BACKEND'] = 'tests._bench_gpu_gloo'
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    r = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2',
         '--steps', '12', '--warmup', '4', '--equil', '100', '--pre-equil', '500',
         '--c4-bosons', '128', '--c4-walkers', '16384', '--chains', '4096',
         '--rebalance-every', '4', '--launch-timeout', '500', '--no-checks'],
        env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['scaling'] == 'strong'
    assert out['backend'].startswith('hip + gloo')
    ex = out['extra']
    wu, tm = ex['rebalance_checks']['warmup'], ex['rebalance_checks']['timed']
    # 8192 +- 3 %: 245 walkers moved in the warm-up (sent + received: 490)
    assert max(wu['counts_before']) - min(wu['counts_before']) > 400
    assert max(wu['counts_after']) - min(wu['counts_after']) <= 1
    assert sum(wu['counts_before']) == sum(wu['counts_after'])
    assert sum(tm['counts_before']) == sum(tm['counts_after'])
    assert ex['walkers_moved_warmup_all_ranks'] >= 400
    assert ex['walkers_moved_all_ranks'] > 0
    assert ex['ref_energy_identical_on_all_ranks'] is True
    # (a short run from a seed ensemble of 500 steps + 100 steps: the result
    # windows of the real benchmark are switched off, a sanity window here)
    assert 14.5 < ex['dmc_energy_per_particle'] < 16.5
    assert 0.9 * 16384 < ex['mean_walkers'] < 1.1 * 16384
    assert ex['phases']['evolve_kernel_ms_per_step'] > 0
    # the weak-scaled VMC extra ran on both ranks as well
    assert ex['vmc_weak']['value'] > 0
    # (500 seed steps from one particle per well: E/N still below 15)
    assert 14.5 < ex['vmc_weak']['energy_per_particle'] < 16.5
    # both curves under the same keys, and their same-run 1-GPU references
    # (rank 0 alone: the whole population of 16384 walkers / its 4096 chains)
    assert ex['curves'] == {'vmc_n64_weak': ex['vmc_weak']['value'],
                            'dmc_n128_strong': out['value']}
    for key, curve in (('strong_scaling', 'dmc_n128_strong'),
                       ('weak_scaling', 'vmc_n64_weak')):
        sc = ex[key]
        assert sc['curve'] == curve and sc['n_gpus'] == 2
        assert sc['ref_1gpu'] > 0 and 'same run' in sc['ref_measured']
        assert sc['efficiency'] == pytest.approx(
            ex['curves'][curve] / sc['ref_1gpu'] / 2)
    assert 0.9 * 16384 < ex['strong_scaling']['ref_detail']['mean_walkers'] \
        < 1.1 * 16384


def test_distributed_vmc_reduces_block_sums_on_the_device():
    """`DistributedVmc.run_block` on a GPU ensemble sums the per-chain block sums
    where they are (it used to download 24 bytes per chain); the
    statistics must be those of the per-chain arrays of an identical run."""
    import types
    from phd_qmclib_amd.dist import DistributedVmc
    from phd_qmclib_amd.engine import ModelEngine, VmcEnsemble
    n, W = 64, 3000
    eng = ModelEngine(box(n).cfc_spec)
    pos = n * np.random.RandomState(12).random_sample((W, n))

    def make(first_chain):
        v = VmcEnsemble(eng, W, 0.125, rng_seed=8, chain0=first_chain)
        v.set_state(pos)
        return types.SimpleNamespace(ensemble=v)

    dv = DistributedVmc(make, W, device='cuda')
    ref = make(0).ensemble
    for _ in range(2):
        got = dv.run_block(24)
        out = ref.run_block(24)
        tot = float(W * 24)
        assert got['num_samples'] == tot
        assert got['energy_mean'] == pytest.approx(out['sum_energy'].sum() / tot,
                                                   rel=1e-12)
        assert got['energy2_mean'] == pytest.approx(
            out['sum_energy2'].sum() / tot, rel=1e-12)
        assert got['accept_rate'] == out['num_accepted'].sum() / tot
    dv.sampling.ensemble.close()
    ref.close()
    eng.close()
