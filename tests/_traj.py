"""Helpers of the trajectory tests (device chains against oracle chains on the
same Philox streams).

A device chain and its oracle chain evaluate log|psi| with different summation
orders, so the two values differ at rounding level (<= 2e-11 max(1, |x|), the
suite's tolerance; ~1e-13 in practice).  The Metropolis test
`log psi' > 0.5 ln u + log psi` (qmc_base/vmc.py:636) can therefore come out
differently only when its margin is that small.  P(|margin| < eps) <= 4 eps per
step (the density of 0.5 ln u is <= 2), i.e. ~1e-7 per step for the 1e-8 bound
used here: a flip is a once-in-a-million-steps event, and anything else that
separates the two sides is a bug.  So: every chain whose accept / reject
sequence differs from the oracle's is DIAGNOSED -- the first differing step is
located and the oracle's margin there must be below FLIP_MARGIN -- and at most
one such chain per test is tolerated.
"""
import numpy as np

FLIP_MARGIN = 1e-8


def first_difference(a, b):
    """Index of the first element where the boolean series differ, or None."""
    a, b = np.asarray(a, dtype=bool), np.asarray(b, dtype=bool)
    d = np.nonzero(a != b)[0]
    return int(d[0]) if d.size else None


def vmc_margin(oracle, m, pos0, spread, seed, chain, t):
    """The oracle's Metropolis margin log psi' - (0.5 ln u + log psi) at yield
    `t` (t >= 1; yield 0 is the initial state) of the chain started at pos0:
    the chain is replayed to yield t - 1, the proposal of the next step is
    rebuilt from the shared Philox2x32 move stream (word 0 of particle i's
    block moves it, the second words of particles 0 and 1 are the accept
    draw: oracle/qmc_oracle.c, orc_vmc_move_block)."""
    assert t >= 1
    ch = oracle.VmcChain(m, pos0, spread, seed=seed, chain=chain)
    ch.run(t)
    step = int(ch.cfg.step0)
    n, L = int(m.boson_number), float(m.supercell_size)
    w = [oracle.vmc_move_block(seed, chain, step, i) for i in range(n)]
    d = np.array([oracle.vmc_move_unit(w0) for w0, _ in w]) * spread
    prop = np.mod(ch.pos + d, L)
    wf_new = oracle.wf_abs_log(m, prop)
    ua = oracle.vmc_accept_uniform(w[0][1], w[min(1, n - 1)][1])
    return float(wf_new - (0.5 * np.log(ua) + ch.wf[0]))


def explain_flips(oracle, m, pos0, spread, seed, stat_dev, stat_orc,
                  chain0=0, max_flips=1):
    """stat_dev / stat_orc: [nyield, W] accept series of device and oracle.
    -> boolean mask [W] of the chains whose series agree.  Every other chain
    must be explained by a Metropolis margin below FLIP_MARGIN at its first
    differing step, and there may be at most `max_flips` of them."""
    stat_dev = np.asarray(stat_dev, dtype=bool)
    stat_orc = np.asarray(stat_orc, dtype=bool)
    W = stat_dev.shape[1]
    same = np.ones(W, dtype=bool)
    notes = []
    for c in range(W):
        t = first_difference(stat_dev[:, c], stat_orc[:, c])
        if t is None:
            continue
        same[c] = False
        assert t >= 1, (c, 'the initial yield is always ACCEPTED')
        mg = vmc_margin(oracle, m, pos0[c], spread, seed, chain0 + c, t)
        notes.append((c, t, mg))
        assert abs(mg) < FLIP_MARGIN, \
            f'chain {c} leaves the oracle at yield {t} with margin {mg:.3e}: ' \
            f'not a rounding-level flip'
    assert len(notes) <= max_flips, notes
    if notes:
        print('marginal Metropolis flips (chain, yield, margin):', notes)
    return same
