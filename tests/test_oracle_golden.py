"""Pins the CPU oracle (oracle/qmc_oracle.c) to the reference: every check
compares against vectors the reference's own function bodies produced
(tests/golden/, generator: oracle/refgen/gen_golden.py)."""
import os

import numpy as np
import pytest

from .conftest import GOLDEN, oracle_model

TAGS = ['box8', 'box16', 'box64', 'box128', 'box512', 'free16', 'deep100',
        'deep16', 'ideal16', 'defect24', 'odd24', 'box37', 'box48', 'box100',
        'box126']


@pytest.mark.parametrize('tag', TAGS)
def test_kernels_bit_exact(oracle, golden_params, golden_kernels, tag):
    """wf_abs_log / energy / ith_energy_and_drift: same libm, same operation
    order => bit-identical to the reference (jastrow/model.py:298-366,793-854)."""
    m = oracle_model(oracle, golden_params, tag)
    pos = golden_kernels[tag + '/pos']
    for k in range(len(pos)):
        wf = oracle.wf_abs_log(m, pos[k])
        e, ie, fd = oracle.energy_drift(m, pos[k])
        assert wf == golden_kernels[tag + '/wf_abs_log'][k]
        assert e == golden_kernels[tag + '/energy'][k]
        assert np.array_equal(ie, golden_kernels[tag + '/ith_energy'][k])
        assert np.array_equal(fd, golden_kernels[tag + '/ith_drift'][k])


@pytest.mark.parametrize('tag', ['box8', 'box16', 'free16', 'deep16',
                                 'defect24', 'box64', 'odd24'])
def test_vmc_tape_replay(oracle, golden_params, golden_vmc_tape, tag):
    """Replays the reference's recorded rand() stream through the oracle's
    Metropolis chain: per-step log-psi, move status and energy must match
    (qmc_base/vmc.py:624-646, 725-768; jastrow/vmc.py:237-262)."""
    g = golden_vmc_tape
    m = oracle_model(oracle, golden_params, tag)
    ch = oracle.VmcChain(m, g[tag + '/ini_pos'], float(g[tag + '/move_spread']))
    assert ch.wf[0] == float(g[tag + '/ini_wf_abs_log'])
    nblocks, ns = g[tag + '/wf_abs_log'].shape
    n = m.boson_number
    tape = g[tag + '/uniform']
    off = 0
    for b in range(nblocks):
        real = ns - (1 if b == 0 else 0)
        wf, en, st, acc = ch.run(ns, tape[off:off + real * (n + 1)])
        off += real * (n + 1)
        assert np.array_equal(st, g[tag + '/move_stat'][b])
        assert np.array_equal(wf, g[tag + '/wf_abs_log'][b])
        assert np.array_equal(en, g[tag + '/energy'][b])
        assert acc / ns == g[tag + '/accept_rate'][b]
    assert off == tape.size
    assert np.array_equal(ch.pos, g[tag + '/last_pos'])


@pytest.mark.parametrize('tag', ['box8', 'box16', 'free16', 'cap8', 'box64',
                                 'odd24'])
def test_dmc_tape_replay(oracle, golden_params, golden_dmc_tape, tag):
    """Replays the reference's rand()/normal() streams through the oracle's
    DMC generator: branching table, per-step scalars, yielded walkers
    (qmc_base/dmc.py:622-653, 739-785; jastrow/dmc.py:758-825, 892-942)."""
    g = golden_dmc_tape
    stag = 'box8' if tag == 'cap8' else tag
    m = oracle_model(oracle, golden_params, stag)
    dt, target, maxw, kappa, steps, n_ini = g[tag + '/cfg']
    target, maxw, steps, n_ini = int(target), int(maxw), int(steps), int(n_ini)
    ens = oracle.DmcEnsemble(m, g[tag + '/ini_pos'], dt, maxw, target, kappa)
    assert np.array_equal(ens.ini_energy[:n_ini], g[tag + '/ini_energy'])
    assert np.array_equal(ens.ini_confs[:n_ini, 1], g[tag + '/ini_drift'])
    assert ens.st.ref_energy == float(g[tag + '/ini_ref_energy'])
    u, gg = g[tag + '/uniform'], g[tag + '/normal']
    uo = go = 0
    hit_cap = False
    for t in range(steps):
        nu, nn = int(g[tag + '/n_uniform'][t]), int(g[tag + '/n_normal'][t])
        out = ens.step(u[uo:uo + nu + 64], gg[go:go + nn])
        assert out.n_uniform == nu and out.n_normal == nn
        uo += nu
        go += nn
        nw = int(g[tag + '/num_walkers'][t])
        hit_cap |= nw == maxw
        assert out.num_walkers == nw
        assert np.array_equal(ens.cloning_ref[:nw],
                              g[tag + '/cloning_ref'][t, :nw])
        assert out.energy == g[tag + '/energy'][t]
        assert out.weight == g[tag + '/weight'][t]
        assert out.ref_energy == g[tag + '/ref_energy'][t]
        assert out.accum_energy == g[tag + '/accum_energy'][t]
        assert np.array_equal(ens.confs[:nw], g[tag + '/confs'][t, :nw])
        assert np.array_equal(ens.energy[:nw],
                              g[tag + '/walker_energy'][t, :nw])
    assert uo == u.size and go == gg.size
    if tag == 'cap8':
        assert hit_cap


def test_np_sum_matches_numpy(oracle):
    rng = np.random.RandomState(3)
    for n in (1, 2, 7, 8, 9, 24, 100, 128, 129, 300, 1000, 5000):
        a = rng.normal(size=n) * 1e3
        assert oracle.np_sum(a) == a.sum()


def test_philox_known_answer(oracle):
    """Random123 known-answer vectors for Philox4x32-10
    (counter, key) = (0,0) and (ff..f, ff..f)."""
    import ctypes as C
    L = oracle.lib()
    u = oracle.philox_uniform2(0, 0, 0, 0, 0)
    # KAT: philox4x32_10 ctr=0 key=0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8
    def u53(hi, lo):
        return float((((hi >> 5) << 26) | (lo >> 6))) / 9007199254740992.0
    assert u[0] == u53(0x6627e8d5, 0xe169c58d)
    assert u[1] == u53(0xbc57ac4c, 0x9b00dbd8)
    u = oracle.philox_uniform2(0xffffffffffffffff, 0xffffffff, 0xffffffff,
                               0xffffffff, 0xffffffff)
    # KAT: ctr=ff.. key=ff.. -> 408f276d 41c83b0e a20bc7c6 6d5451fd
    assert u[0] == u53(0x408f276d, 0x41c83b0e)
    assert u[1] == u53(0xa20bc7c6, 0x6d5451fd)


def test_philox2x32_known_answer(oracle):
    """Random123 known-answer vectors for Philox2x32-10 (kat_vectors:
    counter, key -> block), the generator of the VMC move stream, and the
    stream's keying: distinct counters for distinct (chain, step, particle),
    the exact displacement / accept-draw arithmetic."""
    assert oracle.philox2x32(0, 0, 0) == (0xff1dae59, 0x6cd10df2)
    assert oracle.philox2x32(0xffffffff, 0xffffffff, 0xffffffff) == \
        (0x2c3f628b, 0xab4fd7ad)
    assert oracle.philox2x32(0x243f6a88, 0x85a308d3, 0x13198a2e) == \
        (0xdd7ce038, 0xf62a4c12)
    # keying (oracle/qmc_oracle.c: orc_vmc_move_block)
    seed = 0x0123456789abcdef
    key = ((seed & 0xffffffff) ^ ((seed >> 32) * 0x85EBCA6B)) & 0xffffffff
    slot, step, idx = 0x2345678, 0x1abcdef, 0x2a5
    c0 = ((step & 0x3ffffff) << 6) | ((slot >> 22) & 0x3f)
    c1 = ((slot & 0x3fffff) << 10) | idx
    assert oracle.vmc_move_block(seed, slot, step, idx) == \
        oracle.philox2x32(c0, c1, key)
    seen = set()
    for sl in (0, 1, (1 << 22) - 1, 1 << 22, (1 << 28) - 1):
        for st in (0, 1, (1 << 26) - 1):
            for i in (0, 1, 511, 1023):
                seen.add(oracle.vmc_move_block(7, sl, st, i))
    assert len(seen) == 5 * 3 * 4
    # bits beyond the counter's ranges move into the key
    assert oracle.vmc_move_block(7, 1 << 28, 0, 0) != \
        oracle.vmc_move_block(7, 0, 0, 0)
    assert oracle.vmc_move_block(7, 0, 1 << 26, 0) != \
        oracle.vmc_move_block(7, 0, 0, 0)
    assert oracle.vmc_move_unit(0) == 2.0 ** -33 - 0.5
    assert oracle.vmc_move_unit(0xffffffff) == 0.5 - 2.0 ** -33
    assert oracle.vmc_move_unit(0x80000000) == 2.0 ** -33
    assert oracle.vmc_accept_uniform(0xffffffff, 0xffffffff) == \
        1.0 - 2.0 ** -53
    assert oracle.vmc_accept_uniform(0x12345678, 0x9abcdef0) == float(
        ((0x12345678 >> 5) << 26) | (0x9abcdef0 >> 6)) / 2.0 ** 53


def test_dmc_normal_stream_definition(oracle):
    """The DMC diffusion stream (oracle/qmc_oracle.c: orc_dmc_normal2): one
    Philox2x32-10 block per (walker slot, pair of time steps, particle) under
    the seed's key + 0x27D4EB2F, 32-bit uniforms (w + 1/2) 2^-32, Box-Muller:
    cosine branch at step 2m, sine branch at 2m + 1.  Restated here from the
    block function pinned by the Random123 vectors above; and the moments of
    40 000 draws."""
    import math
    seed = 0x0123456789abcdef
    key = ((seed & 0xffffffff) ^ ((seed >> 32) * 0x85EBCA6B)) & 0xffffffff
    key = (key + 0x27D4EB2F) & 0xffffffff
    for slot, step, idx in ((0x2345678, 0x1abcde, 0x2a5), (0, 0, 0), (5, 7, 63)):
        c0 = (((step >> 1) & 0x3ffffff) << 6) | ((slot >> 22) & 0x3f)
        c1 = ((slot & 0x3fffff) << 10) | idx
        w0, w1 = oracle.philox2x32(c0, c1, key)
        u0, u1 = (w0 + 0.5) / 2.0 ** 32, (w1 + 0.5) / 2.0 ** 32
        r = math.sqrt(-2.0 * math.log(u0))
        g = (r * math.cos(2 * math.pi * u1), r * math.sin(2 * math.pi * u1))
        got = oracle.dmc_normal(seed, slot, step, idx)
        assert abs(got - g[step & 1]) <= 1e-15 * max(1.0, abs(got))
        # the other step of the pair takes the other branch of the same block
        got2 = oracle.dmc_normal(seed, slot, step ^ 1, idx)
        assert abs(got2 - g[(step ^ 1) & 1]) <= 1e-15 * max(1.0, abs(got2))
    # another walker, another particle, another step pair: other blocks
    base = oracle.dmc_normal(7, 3, 10, 5)
    assert oracle.dmc_normal(7, 4, 10, 5) != base
    assert oracle.dmc_normal(7, 3, 12, 5) != base
    assert oracle.dmc_normal(7, 3, 10, 6) != base
    assert oracle.dmc_normal(7, 3 + (1 << 28), 10, 5) != base
    g = np.array([oracle.dmc_normal(11, s, t, i) for s in range(50)
                  for t in range(20) for i in range(40)])
    n = g.size
    assert abs(g.mean()) < 4.0 / math.sqrt(n)
    assert abs(g.var() - 1.0) < 4.0 * math.sqrt(2.0 / n)
    assert abs((g ** 4).mean() - 3.0) < 4.0 * math.sqrt(96.0 / n)
    assert np.abs(g).max() < 6.77


def test_vmc_ndf_tape_replay(oracle, golden_params):
    """Gaussian-proposal VMC (qmc_base/vmc_ndf.py:43-59, mrbp_qmc/vmc_ndf.py):
    the oracle's chain on the reference's recorded normal()/rand() streams --
    pins the oracle path the device's `NDFSampling` is compared with."""
    g = np.load(os.path.join(GOLDEN, 'vmc_extra.npz'), allow_pickle=False)
    m = oracle_model(oracle, golden_params, 'box8')
    sigma = float(np.sqrt(g['ndf8/time_step']))
    ch = oracle.VmcChain(m, g['ndf8/ini_pos'], sigma, gaussian=True)
    tape = g['ndf8/tape']
    nblocks, ns = g['ndf8/wf_abs_log'].shape
    off = 0
    for b in range(nblocks):
        real = ns - (1 if b == 0 else 0)
        wf, en, st, acc = ch.run(ns, tape[off:off + real].ravel())
        off += real
        assert np.array_equal(st, g['ndf8/move_stat'][b])
        assert np.array_equal(wf, g['ndf8/wf_abs_log'][b])
        assert np.array_equal(en, g['ndf8/energy'][b])
        assert acc / ns == g['ndf8/accept_rate'][b]
    assert off == len(tape)
    assert np.array_equal(ch.pos, g['ndf8/last_pos'])
