"""Parity of the HIP engine (through the C-ABI) against the reference's golden
vectors and against the CPU oracle on seeded inputs.  All tests need a GPU.

Tolerances: the device code evaluates the same formulas with a different (but
mathematically identical) arrangement -- angle-difference identities instead
of a tan() per pair, N(N-1)/2 unordered pairs instead of N(N-1) ordered ones
-- so deterministic quantities agree to rounding, not bit for bit:
|delta| <= 2e-11 * max(1, |x|) for energies / drifts / log-psi.  Discrete
quantities (move status, clone counts, cloning table, population size) must
match exactly.
"""
import numpy as np
import pytest

from .conftest import oracle_model

pytestmark = pytest.mark.gpu

RTOL = 2e-11
ALL_TAGS = ['box8', 'box16', 'box64', 'box128', 'box512', 'free16', 'deep100',
            'deep16', 'ideal16', 'defect24', 'odd24', 'box37', 'box48', 'box100',
        'box126']


def close(a, b, rtol=RTOL):
    a, b = np.asarray(a), np.asarray(b)
    return np.all(np.abs(a - b) <= rtol * np.maximum(1.0, np.abs(b)))


def worst(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def spec_from_golden(golden_params, tag):
    from phd_qmclib_amd.mrbp_qmc import Spec
    return Spec(**golden_params[tag]['spec'])


@pytest.fixture(scope='module')
def engines(golden_params):
    from phd_qmclib_amd.engine import ModelEngine
    cache = {}

    def get(tag):
        if tag not in cache:
            cache[tag] = ModelEngine(spec_from_golden(golden_params, tag).cfc_spec)
        return cache[tag]
    yield get
    for e in cache.values():
        e.close()


@pytest.mark.parametrize('tag', ALL_TAGS)
def test_evaluate_vs_reference_golden(engines, golden_kernels, tag):
    """wf_abs_log / energy / ith_energy_and_drift on the fixed configurations
    (random, regular, near-contact, wrap-edge) of every golden spec."""
    eng = engines(tag)
    out = eng.evaluate(golden_kernels[tag + '/pos'])
    for name, got in [('wf_abs_log', out.wf_abs_log), ('energy', out.energy),
                      ('ith_energy', out.ith_energy),
                      ('ith_drift', out.drift)]:
        ref = golden_kernels[tag + '/' + name]
        assert close(got, ref), (tag, name, worst(got, ref))


@pytest.mark.parametrize('tag,nconf', [('box16', 300), ('box64', 200),
                                       ('box128', 64), ('box512', 6),
                                       ('deep100', 40), ('defect24', 100),
                                       ('odd24', 100), ('free16', 100)])
def test_evaluate_vs_oracle_random(engines, oracle, golden_params, tag, nconf):
    """Seeded random configurations; every fourth one with particles up to
    two box lengths outside [0, L) on either side (the reference's functions
    are periodic through their minimum-image distances)."""
    m = oracle_model(oracle, golden_params, tag)
    rng = np.random.RandomState(sum(map(ord, tag)))
    L, n = m.supercell_size, m.boson_number
    pos = L * rng.random_sample((nconf, n))
    shift = L * rng.randint(-2, 3, size=(nconf, n))
    shift[np.arange(nconf) % 4 != 0] = 0
    pos = pos + shift
    wf, en, ith, dr = oracle.evaluate_set(m, pos)
    out = engines(tag).evaluate(pos)
    assert close(out.wf_abs_log, wf), worst(out.wf_abs_log, wf)
    assert close(out.energy, en), worst(out.energy, en)
    assert close(out.ith_energy, ith), worst(out.ith_energy, ith)
    assert close(out.drift, dr), worst(out.drift, dr)


def test_evaluate_ragged_batch(engines, oracle, golden_params):
    """Batch sizes that do not fill a workgroup, and a batch of one."""
    m = oracle_model(oracle, golden_params, 'box16')
    rng = np.random.RandomState(5)
    for W in (1, 3, 15, 17, 65):
        pos = 16 * rng.random_sample((W, 16))
        _, en, _, _ = oracle.evaluate_set(m, pos)
        assert close(engines('box16').evaluate(pos).energy, en)


@pytest.mark.parametrize('tag', ['box8', 'box16', 'free16', 'deep16',
                                 'defect24', 'box64', 'odd24'])
def test_vmc_tape_replay(engines, golden_vmc_tape, tag):
    """Replay of the reference's recorded rand() stream: identical accept /
    reject sequence, log-psi and energy series, block by block."""
    from phd_qmclib_amd.engine import VmcEnsemble
    g = golden_vmc_tape
    eng = engines(tag)
    n = eng.num_particles
    nblocks, ns = g[tag + '/wf_abs_log'].shape
    tape = g[tag + '/uniform'].reshape(1, -1, n + 1)
    ens = VmcEnsemble(eng, 1, float(g[tag + '/move_spread']), rng_seed=7)
    ens.set_state(g[tag + '/ini_pos'][None, :])
    ens.set_tape(tape)
    for b in range(nblocks):
        out = ens.run_block(ns, series=True)
        assert np.array_equal(out['move_stat'][:, 0], g[tag + '/move_stat'][b])
        assert close(out['wf_abs_log'][:, 0], g[tag + '/wf_abs_log'][b])
        assert close(out['energy'][:, 0], g[tag + '/energy'][b])
        assert out['num_accepted'][0] / ns == g[tag + '/accept_rate'][b]
        assert close(out['sum_energy'][0], g[tag + '/energy'][b].sum(),
                     rtol=1e-12 * ns)
    pos, wf, _ = ens.get_state()
    assert close(pos[0], g[tag + '/last_pos'], rtol=1e-13)
    ens.close()


@pytest.mark.parametrize('tag', ['box8', 'box16', 'free16', 'cap8', 'box64',
                                 'odd24'])
def test_dmc_tape_replay(engines, golden_dmc_tape, tag):
    """Replay of the reference's rand()/normal() streams through the device
    branching scan + propagation: population sizes and cloning tables exact,
    estimators and yielded walkers to rounding (includes the run that is
    truncated at max_num_walkers)."""
    from phd_qmclib_amd.engine import DmcEnsemble
    g = golden_dmc_tape
    stag = 'box8' if tag == 'cap8' else tag
    eng = engines(stag)
    n = eng.num_particles
    dt, target, maxw, kappa, steps, n_ini = g[tag + '/cfg']
    target, maxw, steps, n_ini = int(target), int(maxw), int(steps), int(n_ini)
    ens = DmcEnsemble(eng, dt, maxw, target, kappa, rng_seed=11)
    # build_state keeps the last `target` configurations (mrbp_qmc/dmc.py:290)
    ens.set_state(g[tag + '/ini_pos'][-target:])
    n_ini = min(n_ini, target)
    st0 = ens.get_state()
    assert st0.num_walkers == n_ini
    assert close(st0.energy[:n_ini], g[tag + '/ini_energy'][:n_ini])
    assert close(st0.confs[:n_ini, 1], g[tag + '/ini_drift'][:n_ini])
    assert close(st0.ref_energy, float(g[tag + '/ini_ref_energy']))
    nu, nn = g[tag + '/n_uniform'], g[tag + '/n_normal']
    u_off = np.concatenate([[0], np.cumsum(nu)[:-1]])
    g_off = np.concatenate([[0], np.cumsum(nn)[:-1]])
    ens.set_tape(g[tag + '/uniform'], g[tag + '/normal'], u_off, g_off)
    for t in range(steps):
        ser = ens.run_block(1)
        nw = int(g[tag + '/num_walkers'][t])
        assert int(ser.num_walkers[0]) == nw, t
        st = ens.get_state()
        assert st.num_walkers == nw
        assert np.array_equal(st.cloning_ref[:nw],
                              g[tag + '/cloning_ref'][t, :nw]), t
        assert close(ser.energy[0], g[tag + '/energy'][t]), t
        assert ser.weight[0] == g[tag + '/weight'][t]
        assert close(ser.ref_energy[0], g[tag + '/ref_energy'][t]), t
        assert close(ser.accum_energy[0], g[tag + '/accum_energy'][t]), t
        assert close(st.confs[:nw], g[tag + '/confs'][t, :nw], rtol=1e-10), t
        assert close(st.energy[:nw], g[tag + '/walker_energy'][t, :nw]), t
        assert not st.mask[:nw].any() and st.mask[nw:].all()
    ens.close()


def test_dmc_tape_replay_one_block(engines, golden_dmc_tape):
    """The same replay enqueued as ONE block call (no host sync between time
    steps) gives the same series as step-by-step."""
    from phd_qmclib_amd.engine import DmcEnsemble
    g, tag = golden_dmc_tape, 'box8'
    eng = engines('box8')
    dt, target, maxw, kappa, steps, n_ini = g[tag + '/cfg']
    ens = DmcEnsemble(eng, dt, int(maxw), int(target), kappa, rng_seed=11)
    ens.set_state(g[tag + '/ini_pos'])
    nu, nn = g[tag + '/n_uniform'], g[tag + '/n_normal']
    ens.set_tape(g[tag + '/uniform'], g[tag + '/normal'],
                 np.concatenate([[0], np.cumsum(nu)[:-1]]),
                 np.concatenate([[0], np.cumsum(nn)[:-1]]))
    ser = ens.run_block(int(steps))
    assert np.array_equal(ser.num_walkers.astype(np.int64),
                          g[tag + '/num_walkers'])
    assert close(ser.energy, g[tag + '/energy'])
    assert close(ser.ref_energy, g[tag + '/ref_energy'])
    assert close(ser.accum_energy, g[tag + '/accum_energy'])
    ens.close()


def test_dmc_initial_weights_reach_the_branching(engines, golden_kernels):
    """`set_full_state` with weights that are not 1 (a continued run hands over
    the last state of the previous one, mrbp_qmc/dmc.py:246-262): the state
    reads back as given to the bit, and the first branching takes
    int(w + u) children of every walker (qmc_base/jastrow/dmc.py:860-878),
    zero included, the cap of max_num_walkers respected.  The device keeps
    the LOGARITHMS of the weights (dmc_evolve_kernel): this is the one place
    where a caller's weights are converted."""
    from phd_qmclib_amd.engine import DmcEnsemble
    tag = 'box16'
    eng = engines(tag)
    pos = np.tile(golden_kernels[tag + '/pos'], (2, 1))[:12]
    W, n = pos.shape
    maxw = 40
    d = DmcEnsemble(eng, 1e-300, maxw, W, 0.5, rng_seed=1)
    d.set_state(pos)
    st = d.get_state()
    assert np.array_equal(st.weight[:W], np.ones(W))
    w = np.array([0.0, 0.3, 0.5, 1.0, 1.0 - 2.0 ** -53, 1.7, 2.3, 3.999, 1e-320,
                  0.95, 2.0, 1.25])
    # (w = 1 is exact -- log 1 = 0 -- and it is the weight every state the
    # library itself yields carries; for other weights exp(log w) may sit one
    # ulp from w, so the draws here keep w + u off the integers)
    u = np.array([0.9, 0.69, 0.51, 0.0, 0.25, 0.31, 0.69, 0.0009, 0.999,
                  0.05, 0.999, 0.74])
    d.set_full_state(st.confs[:W], st.energy[:W], w, st.ref_energy)
    st1 = d.get_state()
    assert np.array_equal(st1.weight[:W], w) and not st1.weight[W:].any()
    assert np.array_equal(st1.energy[:W], st.energy[:W])
    kids = (w + u).astype(np.int64)
    nw1 = int(kids.sum())
    assert 0 < nw1 <= maxw
    d.set_tape(u, np.zeros(nw1 * n), [0], [0])
    ser = d.run_block(1)
    assert int(ser.num_walkers[0]) == nw1
    st2 = d.get_state()
    assert np.array_equal(st2.cloning_ref[:nw1], np.repeat(np.arange(W), kids))
    # the children sit where their parents sat, with their parents' energies
    par = st2.cloning_ref[:nw1]
    assert np.abs(st2.confs[:nw1, 0] - st.confs[par, 0]).max() <= 1e-250
    assert close(st2.energy[:nw1], st.energy[par])
    assert np.array_equal(st2.weight[:nw1], np.ones(nw1))
    # total weight of the yield: one per child (unit weights after branching)
    assert ser.weight[0] == nw1
    d.close()


def test_vmc_philox_matches_oracle(engines, oracle, golden_params):
    """Same seed, same counter RNG: device chains and oracle chains follow the
    same trajectories (Philox4x32-10 keyed by (seed; chain, step, particle,
    stream) on both sides).  A chain may leave the oracle only through a
    Metropolis test whose margin is at rounding level, which is checked
    (tests/_traj.py); observed here: 0 of 37 chains."""
    from phd_qmclib_amd.engine import VmcEnsemble
    from ._traj import explain_flips
    tag, W, ns = 'box16', 37, 24
    m = oracle_model(oracle, golden_params, tag)
    rng = np.random.RandomState(17)
    pos0 = 16 * rng.random_sample((W, 16))
    ens = VmcEnsemble(engines(tag), W, 0.125, rng_seed=123456789)
    ens.set_state(pos0)
    out = ens.run_block(ns, series=True)
    out2 = ens.run_block(ns, series=True)
    st_o, en_o, wf_o = [np.zeros((2 * ns, W)) for _ in range(3)]
    for c in range(W):
        ch = oracle.VmcChain(m, pos0[c], 0.125, seed=123456789, chain=c)
        wf, en, st, _ = ch.run(ns)
        wf2, en2, st2, _ = ch.run(ns)
        st_o[:, c], en_o[:, c], wf_o[:, c] = np.r_[st, st2], np.r_[en, en2], \
            np.r_[wf, wf2]
    same = explain_flips(oracle, m, pos0, 0.125, 123456789,
                         np.r_[out['move_stat'], out2['move_stat']], st_o)
    assert close(en_o[:, same], np.r_[out['energy'], out2['energy']][:, same],
                 1e-9)
    assert close(wf_o[:, same],
                 np.r_[out['wf_abs_log'], out2['wf_abs_log']][:, same], 1e-9)
    ens.close()


def test_dmc_philox_matches_oracle(engines, oracle, golden_params):
    from phd_qmclib_amd.engine import DmcEnsemble
    tag = 'box16'
    m = oracle_model(oracle, golden_params, tag)
    rng = np.random.RandomState(23)
    pos0 = 16 * rng.random_sample((200, 16))
    cfg = dict(time_step=1e-3, max_num_walkers=256, target_num_walkers=200)
    ens = DmcEnsemble(engines(tag), num_walkers_control_factor=0.5,
                      rng_seed=424242, **cfg)
    ens.set_state(pos0)
    orc = oracle.DmcEnsemble(m, pos0, 1e-3, 256, 200, 0.5, seed=424242)
    ser = ens.run_block(12)
    for t in range(12):
        o = orc.step()
        assert int(ser.num_walkers[t]) == o.num_walkers, t
        assert close(ser.energy[t], o.energy, 1e-9), t
        assert close(ser.ref_energy[t], o.ref_energy, 1e-9), t
    st = ens.get_state()
    assert np.array_equal(st.cloning_ref[:st.num_walkers],
                          orc.cloning_ref[:st.num_walkers])
    ens.close()


def test_edge_cases_and_errors(engines, golden_params):
    """Empty batch, single chain, size limits and the error channel of the
    C-ABI (status code + qmc_last_error -> QmcError)."""
    from phd_qmclib_amd.engine import (DmcEnsemble, ModelEngine, QmcError,
                                       VmcEnsemble)
    from phd_qmclib_amd.mrbp_qmc import Spec
    eng = engines('box16')
    out = eng.evaluate(np.zeros((0, 16)))
    assert out.energy.shape == (0,)
    with pytest.raises(ValueError):
        eng.evaluate(np.zeros((3, 15)))
    big = dict(golden_params['box16']['spec'], boson_number=513,
               supercell_size=513, tbf_contact_cutoff=100)
    with pytest.raises(QmcError, match='boson_number'):
        ModelEngine(Spec(**big).cfc_spec)
    with pytest.raises(QmcError):
        VmcEnsemble(eng, 0, 0.1, 1)
    v = VmcEnsemble(eng, 1, 0.125, rng_seed=1)
    with pytest.raises(ValueError):
        v.set_state(np.zeros((2, 16)))
    v.set_state(8 * np.ones((1, 16)) + np.arange(16) * 0.4)
    with pytest.raises(QmcError):
        v.run_block(0)
    assert v.run_block(3)['num_accepted'][0] >= 1   # the initial yield counts
    v.close()
    with pytest.raises(QmcError):
        DmcEnsemble(eng, -1.0, 16, 8, 0.5, 1)
    d = DmcEnsemble(eng, 1e-3, 16, 8, 0.5, 1)
    with pytest.raises(QmcError, match='out of range'):
        d.set_state(np.zeros((17, 16)))
    d.close()


@pytest.mark.parametrize('tag', ['ssf_mixed', 'ssf_pure', 'ssf_pure_full',
                                 'dens_mixed', 'dens_pure', 'both'])
def test_dmc_estimators_tape_replay(engines, oracle, golden_params, tag):
    """S(k) and density estimators (mixed / pure) of the device against the
    reference's `Sampling.blocks` output on a replayed run, burned block and
    per-block resets included (tests/golden/dmc_est.npz)."""
    import os
    from phd_qmclib_amd.engine import DmcEnsemble
    from .conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, 'dmc_est.npz'), allow_pickle=False)
    eng = engines('box8')
    dt, target, maxw, kappa, nts, nblocks, burn = g[tag + '/cfg']
    target, maxw, nts, nblocks, burn = map(int, (target, maxw, nts, nblocks,
                                                 burn))
    # per-step draw counts from the (pinned) oracle replay
    m = oracle_model(oracle, golden_params, 'box8')
    orc = oracle.DmcEnsemble(m, g[tag + '/ini_pos'], dt, maxw, target, kappa)
    u, gg = g[tag + '/uniform'], g[tag + '/normal']
    u_off, g_off, uo, go = [], [], 0, 0
    for _ in range(nts * nblocks):
        u_off.append(uo); g_off.append(go)
        o = orc.step(np.r_[u[uo:], np.zeros(64)], gg[go:])
        uo += o.n_uniform; go += o.n_normal
    ens = DmcEnsemble(eng, dt, maxw, target, kappa, rng_seed=3)
    ens.set_state(g[tag + '/ini_pos'][-target:])
    ens.set_tape(u, gg, u_off, g_off)
    kw = {}
    if tag + '/ssf_cfg' in g:
        c = g[tag + '/ssf_cfg']
        kw.update(num_modes=int(c[0]), ssf_pure=bool(c[1]), ssf_pfw=int(c[2]))
    if tag + '/dens_cfg' in g:
        c = g[tag + '/dens_cfg']
        kw.update(num_bins=int(c[0]), dens_pure=bool(c[1]), dens_pfw=int(c[2]))
    ens.set_estimators(**kw)
    for b in range(nblocks):
        ser, ssf, dens = ens.run_block_est(nts, b >= burn)
        assert np.array_equal(ser.num_walkers.astype(np.int64),
                              g[tag + '/num_walkers'][b])
        if ssf is not None:
            ref = g[tag + '/iter_ssf'][b]
            assert np.allclose(ssf, ref, rtol=1e-10, atol=1e-9), (b, worst(ssf, ref))
        if dens is not None:
            assert np.array_equal(dens, g[tag + '/iter_density'][b]), b
    ens.close()


@pytest.mark.parametrize('cutoff', [1.36, 2.7, 6.5])
def test_trajectories_at_other_cutoffs(oracle, cutoff):
    """Contact cutoffs other than L/4: k2 L lands in other quadrants
    (sin(k2 L) < 0 for the first two), which the wrap correction of the
    short-range branch must honour.  VMC and DMC vs the oracle on the same
    Philox streams."""
    from math import pi
    from phd_qmclib_amd import mrbp_qmc
    from phd_qmclib_amd.engine import DmcEnsemble, ModelEngine, VmcEnsemble
    spec = mrbp_qmc.Spec(lattice_depth=5 * pi ** 2, lattice_ratio=1,
                         interaction_strength=2, boson_number=16,
                         supercell_size=16, tbf_contact_cutoff=cutoff)
    m = oracle.model_from_cfc(spec.cfc_spec)
    eng = ModelEngine(spec.cfc_spec)
    rng = np.random.RandomState(31)
    W, ns = 24, 16
    pos0 = 16 * rng.random_sample((W, 16))
    ens = VmcEnsemble(eng, W, 0.125, rng_seed=99)
    ens.set_state(pos0)
    out = ens.run_block(ns, series=True)
    st_o, en_o, wf_o = [np.zeros((ns, W)) for _ in range(3)]
    for c in range(W):
        wf_o[:, c], en_o[:, c], st_o[:, c], _ = oracle.VmcChain(
            m, pos0[c], 0.125, seed=99, chain=c).run(ns)
    # (a chain may leave the oracle only through a rounding-level Metropolis
    # margin, checked in tests/_traj.py; observed: none)
    from ._traj import explain_flips
    same = explain_flips(oracle, m, pos0, 0.125, 99, out['move_stat'], st_o)
    assert close(en_o[:, same], out['energy'][:, same], 1e-9)
    assert close(wf_o[:, same], out['wf_abs_log'][:, same], 1e-9)
    ens.close()
    d = DmcEnsemble(eng, 1e-3, 64, 48, 0.5, rng_seed=77)
    d.set_state(pos0)
    orc = oracle.DmcEnsemble(m, pos0, 1e-3, 64, 48, 0.5, seed=77)
    ser = d.run_block(10)
    for t in range(10):
        o = orc.step()
        assert int(ser.num_walkers[t]) == o.num_walkers, t
        assert close(ser.energy[t], o.energy, 1e-9), t
    d.close()
    eng.close()


@pytest.mark.parametrize('n,modes', [(16, 100), (16, 256), (37, 64), (70, 64),
                                     (70, 130), (9, 3)])
def test_ssf_matrix_core_kernel_vs_numpy(n, modes):
    """S(k) parts of one time step against a direct numpy evaluation on the
    yielded walkers: both tilings of the MFMA kernel (<= 64 modes: one packed
    tile; <= 256: four tiles), particle counts that are not multiples of the
    MFMA K = 4 nor of the 64-particle chunk."""
    from math import pi
    from phd_qmclib_amd import mrbp_qmc
    from phd_qmclib_amd.engine import DmcEnsemble, ModelEngine
    spec = mrbp_qmc.Spec(lattice_depth=5 * pi ** 2, lattice_ratio=1,
                         interaction_strength=2, boson_number=n,
                         supercell_size=n, tbf_contact_cutoff=0.25 * n)
    eng = ModelEngine(spec.cfc_spec)
    rng = np.random.RandomState(n + modes)
    W = 300
    ens = DmcEnsemble(eng, 1e-3, 384, W, 0.5, rng_seed=5)
    ens.set_state(n * rng.random_sample((W, n)))
    ens.set_estimators(num_modes=modes)
    ser, ssf, _ = ens.run_block_est(2, True)
    st = ens.get_state()
    nw = st.num_walkers
    assert nw == int(ser.num_walkers[-1])
    z = st.confs[:nw, 0, :]
    k = 2 * pi * np.arange(modes) / n
    ph = np.exp(1j * k[None, :, None] * z[:, None, :]).sum(axis=2)   # [w, m]
    ref = np.stack([(np.abs(ph) ** 2).sum(0), ph.real.sum(0), ph.imag.sum(0)],
                   axis=1)
    scale = np.abs(ref).max()
    assert np.abs(ssf[-1] - ref).max() <= 1e-11 * scale, \
        np.abs(ssf[-1] - ref).max() / scale
    ens.close()
    eng.close()


@pytest.mark.parametrize('n,bins', [(16, 256), (70, 200), (300, 33), (9, 5)])
def test_density_kernel_vs_numpy(n, bins):
    """Density histogram of the first estimator step against numpy on the
    yielded walkers, for bin counts above the 64 lanes of a wavefront and
    particle counts above one chunk."""
    from math import pi
    from phd_qmclib_amd import mrbp_qmc
    from phd_qmclib_amd.engine import DmcEnsemble, ModelEngine
    spec = mrbp_qmc.Spec(lattice_depth=5 * pi ** 2, lattice_ratio=1,
                         interaction_strength=2, boson_number=n,
                         supercell_size=n, tbf_contact_cutoff=0.25 * n)
    eng = ModelEngine(spec.cfc_spec)
    rng = np.random.RandomState(n * bins)
    W = 200
    ens = DmcEnsemble(eng, 1e-3, 256, W, 0.5, rng_seed=2)
    ens.set_state(n * rng.random_sample((W, n)))
    ens.set_estimators(num_bins=bins)
    ser, _, dens = ens.run_block_est(1, True)
    st = ens.get_state()
    nw = st.num_walkers
    z = st.confs[:nw, 0, :]
    idx = np.clip(np.floor(z / (n / bins)).astype(int), 0, bins - 1)
    ref = np.bincount(idx.ravel(), minlength=bins).astype(float)
    assert dens.shape == (1, bins, 1)
    assert np.array_equal(dens[0, :, 0], ref)
    assert dens.sum() == nw * n
    ens.close()
    eng.close()
