"""What round 1 left unpinned: the device path
against outputs of the REFERENCE itself on recorded random streams
(oracle/refgen/gen_golden.py: `vmc_extra`, `proc`):

  * `Proc.exec` of the VMC and DMC procedures -- block totals and weight
    totals, their means and errors, the default burn-in, with and without
    `keep_iter_data`, with estimators;
  * the Gaussian-proposal sampling `vmc_ndf`;
  * the static structure factor of a single VMC chain, rejected steps and a
    block boundary included;
  * `dmc.Sampling.state_data_blocks` (target overridden by the initial
    population).
"""
import os
import warnings
from itertools import islice
from math import pi, sqrt

import numpy as np
import pytest

from .conftest import GOLDEN

pytestmark = pytest.mark.gpu

RTOL = 2e-11


def close(a, b, rtol=RTOL):
    """(nan matches nan: the error of S(k = 0) is 0 / 0 in the reference.)"""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    both_nan = np.isnan(a) & np.isnan(b)
    return np.all(both_nan |
                  (np.abs(a - b) <= rtol * np.maximum(1.0, np.abs(b))))


def box8():
    from phd_qmclib_amd import mrbp_qmc
    return mrbp_qmc.Spec(lattice_depth=5 * pi ** 2, lattice_ratio=1,
                         interaction_strength=2, boson_number=8,
                         supercell_size=8, tbf_contact_cutoff=2)


@pytest.fixture(scope='module')
def g_extra():
    return np.load(os.path.join(GOLDEN, 'vmc_extra.npz'), allow_pickle=False)


@pytest.fixture(scope='module')
def g_proc():
    return np.load(os.path.join(GOLDEN, 'proc_exec.npz'), allow_pickle=False)


def test_vmc_single_chain_ssf_vs_reference(g_extra):
    """qmc_base/jastrow/vmc.py:304-351 through `Sampling.blocks`: 42 of the 80
    recorded steps are rejections (the reference copies the previous row
    there; the configuration is unchanged, so the recomputed row is the same
    numbers)."""
    from phd_qmclib_amd import mrbp_qmc
    g = g_extra
    spec = box8()
    smp = mrbp_qmc.vmc.Sampling(spec, float(g['ssf8/move_spread']), rng_seed=5,
                                ssf_est_spec=mrbp_qmc.vmc.SSFEstSpec(
                                    int(g['ssf8/num_modes'])))
    smp.set_replay_tape(g['ssf8/uniform'].reshape(-1, 9))
    ini = spec.get_sys_conf_buffer()
    ini[0] = g['ssf8/ini_pos']
    blocks = list(islice(smp.blocks(40, smp.build_state(ini)), 2))
    assert (~g['ssf8/move_stat']).sum() > 30          # rejections are covered
    for b, blk in enumerate(blocks):
        assert np.array_equal(blk.iter_props.move_stat, g['ssf8/move_stat'][b])
        assert close(blk.iter_props.wf_abs_log, g['ssf8/wf_abs_log'][b])
        assert close(blk.iter_props.energy, g['ssf8/energy'][b])
        assert blk.iter_ssf.shape == g['ssf8/iter_ssf'][b].shape
        assert close(blk.iter_ssf, g['ssf8/iter_ssf'][b], rtol=1e-10)


def test_vmc_ndf_vs_reference(g_extra):
    """mrbp_qmc/vmc_ndf.py:23-51, qmc_base/vmc_ndf.py:43-59: normal(0,
    sqrt(time_step)) proposals; the recorded streams of the reference."""
    from phd_qmclib_amd import mrbp_qmc
    g = g_extra
    spec = box8()
    smp = mrbp_qmc.vmc.NDFSampling(spec, time_step=float(g['ndf8/time_step']),
                                   rng_seed=6)
    assert smp._proposal_width() == sqrt(0.01)
    smp.set_replay_tape(g['ndf8/tape'])
    ini = spec.get_sys_conf_buffer()
    ini[0] = g['ndf8/ini_pos']
    blocks = list(islice(smp.blocks(40, smp.build_state(ini)), 2))
    for b, blk in enumerate(blocks):
        assert np.array_equal(blk.iter_props.move_stat, g['ndf8/move_stat'][b])
        assert close(blk.iter_props.wf_abs_log, g['ndf8/wf_abs_log'][b])
        assert close(blk.iter_props.energy, g['ndf8/energy'][b])
        assert blk.accept_rate == g['ndf8/accept_rate'][b]
    assert close(blocks[-1].last_state.sys_conf[0], g['ndf8/last_pos'],
                 rtol=1e-13)


def _check_blocks(blk, g, prefix, rtol=1e-10):
    if prefix + '/totals' in g.files:
        assert close(blk.totals, g[prefix + '/totals'], rtol), prefix
    if prefix + '/weight_totals' in g.files:
        assert close(blk.weight_totals, g[prefix + '/weight_totals'], rtol)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        assert close(blk.mean, g[prefix + '/mean'], rtol), prefix
        assert close(blk.mean_error, g[prefix + '/mean_error'], 1e-8), prefix


@pytest.mark.parametrize('tag', ['vmc', 'vmc_keep'])
def test_vmc_proc_exec_vs_reference(g_proc, tag):
    """qmc_exec/vmc/proc.py:87-250 (driver), :189-217 (reductions): default
    burn-in num_blocks // 8, per-block energy means (or kept series), S(k)
    block means."""
    from phd_qmclib_amd.mrbp_qmc import vmc_exec
    g = g_proc
    spread, nb, ns, keep, nm = g[tag + '/cfg']
    spec = box8()
    proc = vmc_exec.Proc(spec, move_spread=float(spread), rng_seed=8,
                         num_blocks=int(nb), num_steps_block=int(ns),
                         keep_iter_data=bool(keep),
                         ssf_spec=vmc_exec.SSFEstSpec(num_modes=int(nm)))
    assert proc.burn_in_blocks is None
    proc.sampling.set_replay_tape(g[tag + '/uniform'].reshape(-1, 9))
    ini = spec.get_sys_conf_buffer()
    ini[0] = g[tag + '/ini_pos']
    res = proc.exec(vmc_exec.ProcInput(proc.sampling.build_state(ini)))
    _check_blocks(res.data.blocks.energy, g, tag + '/energy')
    _check_blocks(res.data.blocks.ss_factor, g, tag + '/ss_factor')
    assert close(res.state.sys_conf[0], g[tag + '/last_pos'], rtol=1e-13)
    assert close(res.state.wf_abs_log, g[tag + '/last_wf_abs_log'])


@pytest.mark.parametrize('tag', ['dmc', 'dmc_keep', 'dmc_est'])
def test_dmc_proc_exec_vs_reference(g_proc, tag):
    """qmc_exec/dmc/proc.py:136-415 (driver), :308-320 (reductions), :370-415
    (containers): energy / weight / walker-count block totals, the default
    burn-in (num_blocks // 8 with the concrete class), kappa = 0.5 default,
    and with estimators the pure S(k) (last step of a block times the
    population factor) and the mixed density."""
    from phd_qmclib_amd.mrbp_qmc import dmc_exec
    g = g_proc
    dt, maxw, target, kappa, nb, nts, keep = g[tag + '/cfg']
    spec = box8()
    kw = {}
    if tag == 'dmc_est':
        kw = dict(ssf_spec=dmc_exec.SSFEstSpec(num_modes=4, as_pure_est=True),
                  density_spec=dmc_exec.DensityEstSpec(num_bins=8,
                                                       as_pure_est=False))
    proc = dmc_exec.Proc(spec, time_step=float(dt), max_num_walkers=int(maxw),
                         target_num_walkers=int(target), rng_seed=9,
                         num_blocks=int(nb), num_time_steps_block=int(nts),
                         keep_iter_data=bool(keep), **kw)
    assert proc.num_walkers_control_factor == kappa == 0.5
    proc.sampling.set_replay_tape(g[tag + '/uniform'], g[tag + '/normal'],
                                  g[tag + '/n_uniform'], g[tag + '/n_normal'])
    ini_set = np.zeros((len(g[tag + '/ini_pos']), 2, 8))
    ini_set[:, 0, :] = g[tag + '/ini_pos']
    res = proc.exec(dmc_exec.ProcInput(proc.sampling.build_state(ini_set)))
    b = res.data.blocks
    _check_blocks(b.energy, g, tag + '/energy')
    _check_blocks(b.weight, g, tag + '/weight')
    _check_blocks(b.num_walkers, g, tag + '/num_walkers')
    if tag == 'dmc_est':
        _check_blocks(b.ss_factor, g, tag + '/ss_factor')
        _check_blocks(b.density, g, tag + '/density')
    assert res.state.num_walkers == int(g[tag + '/last_num_walkers'])
    assert close(res.state.ref_energy, g[tag + '/last_ref_energy'])


def test_dmc_state_data_blocks_target_quirk(g_proc):
    """qmc_base/dmc.py:974-1070: `state_data_blocks` overrides
    the target population with the initial number of walkers -- so its series
    differ from `blocks` of the same sampling exactly as a sampling whose
    target IS the initial population, and every configuration is kept."""
    from phd_qmclib_amd import mrbp_qmc
    g = g_proc
    spec = box8()
    ini_set = np.zeros((20, 2, 8))
    ini_set[:, 0, :] = g['dmc/ini_pos']

    def smp(target):
        return mrbp_qmc.dmc.Sampling(spec, 1e-3, max_num_walkers=48,
                                     target_num_walkers=target,
                                     num_walkers_control_factor=0.5,
                                     rng_seed=13)
    a = smp(40)                       # target 40, but 20 initial walkers
    st = a.build_state(ini_set)
    assert st.num_walkers == 20
    sdb = next(a.state_data_blocks(st, 12))
    assert sdb.confs.shape == (12, 48, 2, 8)
    ref = smp(20)                     # a sampling whose target is 20
    st20 = ref.build_state(ini_set)
    blk20 = next(ref.blocks(st20, 12, 0))
    blk40 = next(a.blocks(st, 12, 0))
    # same Philox seed, same initial walkers: only the target differs
    assert np.array_equal(sdb.iter_props.num_walkers,
                          blk20.iter_props.num_walkers)
    assert close(sdb.iter_props.ref_energy, blk20.iter_props.ref_energy)
    assert not np.array_equal(blk40.iter_props.ref_energy,
                              blk20.iter_props.ref_energy)
    # every yielded configuration is there: step k's walkers are in the box
    # and the masks follow the population
    for k in range(12):
        nw = int(sdb.iter_props.num_walkers[k])
        assert not sdb.props.mask[k, :nw].any()
        assert sdb.props.mask[k, nw:].all()
        z = sdb.confs[k, :nw, 0]
        assert np.all((z >= 0) & (z < 8))
