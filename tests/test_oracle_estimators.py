"""Pins the oracle's DMC estimators (S(k), density; mixed and pure) to the
reference: tape replay of `Sampling.blocks` runs with estimator specs
(tests/golden/dmc_est.npz)."""
import json
import os

import numpy as np
import pytest

from .conftest import GOLDEN, oracle_model

CASES = ['ssf_mixed', 'ssf_pure', 'ssf_pure_full', 'dens_mixed', 'dens_pure',
         'both']


@pytest.fixture(scope='module')
def golden_est():
    return np.load(os.path.join(GOLDEN, 'dmc_est.npz'), allow_pickle=False)


@pytest.mark.parametrize('tag', CASES)
def test_estimators_tape_replay(oracle, golden_params, golden_est, tag):
    g = golden_est
    m = oracle_model(oracle, golden_params, 'box8')
    dt, target, maxw, kappa, nts, nblocks, burn = g[tag + '/cfg']
    target, maxw, nts, nblocks, burn = map(int, (target, maxw, nts, nblocks,
                                                 burn))
    ens = oracle.DmcEnsemble(m, g[tag + '/ini_pos'], dt, maxw, target, kappa)
    ssf = dens = None
    if tag + '/ssf_cfg' in g:
        c = g[tag + '/ssf_cfg']
        ssf = (int(c[0]), bool(c[1]), int(c[2]))
    if tag + '/dens_cfg' in g:
        c = g[tag + '/dens_cfg']
        dens = (int(c[0]), bool(c[1]), int(c[2]))
    est = oracle.DmcEstimators(m.supercell_size, m.boson_number, maxw, nts,
                               ssf=ssf, dens=dens)
    u, gg = g[tag + '/uniform'], g[tag + '/normal']
    uo = go = 0
    for b in range(nblocks):
        est.reset_block()
        for t in range(nts):
            out = ens.step(np.r_[u[uo:], np.zeros(64)], gg[go:])
            uo += out.n_uniform
            go += out.n_normal
            assert out.num_walkers == g[tag + '/num_walkers'][b, t]
            if b >= burn:
                est.step(t, ens.confs, out.num_walkers, ens.cloning_ref)
        if ssf is not None:
            ref = g[tag + '/iter_ssf'][b]
            assert np.allclose(est.iter_ssf, ref, rtol=1e-12, atol=1e-11), b
            if b < burn:
                assert not ref.any()
        if dens is not None:
            ref = g[tag + '/iter_density'][b]
            assert np.array_equal(est.iter_density, ref), b
    assert uo == u.size and go == gg.size
