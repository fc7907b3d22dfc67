"""Generate the golden vectors under tests/golden/ from the reference itself.

TEST INFRASTRUCTURE ONLY (see harness.py).  Run in the build container:

    MPLBACKEND=Agg /opt/conda/bin/python3.9 oracle/refgen/gen_golden.py [which ...]

`which` in {params, kernels, vmc_tape, dmc_tape, reblock, stats, dmc_est,
wf_opt, sweep}; default all.
Every output is *data* (inputs + the reference's outputs); no reference code
is stored.  numpy seeds are fixed so a re-run reproduces the files.
"""
import json
import os
import sys
from itertools import islice
from math import pi

import harness  # noqa: F401  (must precede phd_qmclib imports)
import numpy as np

from phd_qmclib import mrbp_qmc
from phd_qmclib.mrbp_qmc.model import DIST_REGULAR
from phd_qmclib.stats import reblock as rb
from phd_qmclib.qmc_exec.data import dmc as dmc_data
from phd_qmclib.qmc_exec.data import vmc as vmc_data

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                   '..', '..', 'tests', 'golden')
OUT = os.path.normpath(OUT)
core = mrbp_qmc.model.core_funcs


def box_spec(n, **over):
    """The synthetic "mrbp_qmc box" of SURVEY.md 8(d): unit filling."""
    kw = dict(lattice_depth=5 * pi ** 2, lattice_ratio=1,
              interaction_strength=2, boson_number=n, supercell_size=n,
              tbf_contact_cutoff=0.25 * n)
    kw.update(over)
    return kw


SPECS = {
    'box16': box_spec(16),
    'box64': box_spec(64),
    'box128': box_spec(128),
    'box512': box_spec(512),
    'box8': box_spec(8),
    # tests/mrbp_qmc/test_dmc.py:12-29 (Lieb-Liniger limit with defects off)
    'free16': dict(lattice_depth=0, lattice_ratio=1, interaction_strength=4,
                   boson_number=16, supercell_size=16, tbf_contact_cutoff=4,
                   num_defects=4, defect_magnitude=0),
    # tests/mrbp_qmc/test_model.py:9-14
    'deep100': dict(lattice_depth=100, lattice_ratio=1, interaction_strength=1,
                    boson_number=100, supercell_size=100,
                    tbf_contact_cutoff=25),
    # tests/mrbp_qmc/test_vmc.py:9-14
    'deep16': dict(lattice_depth=100, lattice_ratio=1, interaction_strength=1,
                   boson_number=16, supercell_size=16, tbf_contact_cutoff=4),
    # ideal gas in the lattice (two-body factor switched off)
    'ideal16': dict(lattice_depth=5 * pi ** 2, lattice_ratio=1,
                    interaction_strength=0, boson_number=16, supercell_size=16,
                    tbf_contact_cutoff=4),
    # lattice defects that actually differ from the lattice depth
    'defect24': dict(lattice_depth=5 * pi ** 2, lattice_ratio=0.5,
                     interaction_strength=3, boson_number=20, supercell_size=24,
                     tbf_contact_cutoff=5.5, num_defects=4,
                     defect_magnitude=2 * pi ** 2),
    # non-commensurate filling and ratio, contact cutoff at the box limit
    'odd24': dict(lattice_depth=30.0, lattice_ratio=2.5,
                  interaction_strength=0.7, boson_number=24,
                  supercell_size=17.5, tbf_contact_cutoff=8.75),
    # (round 4, appended so that the random streams of the tags above do not
    # move) sizes of the padded sorted-row shapes of the device engine: rings
    # of 37 / 48 lanes, 50 / 63 lanes with two particles each
    'box37': box_spec(37),
    'box48': box_spec(48),
    'box100': box_spec(100),
    'box126': box_spec(126),
}


def jsonable(nt):
    d = nt._asdict()
    for k, v in d.items():
        if isinstance(v, (np.integer,)):
            d[k] = int(v)
        elif isinstance(v, (np.floating,)):
            d[k] = float(v)
        elif isinstance(v, (np.bool_,)):
            d[k] = bool(v)
    return d


def gen_params():
    out = {}
    for tag, kw in SPECS.items():
        spec = mrbp_qmc.Spec(**kw)
        out[tag] = dict(spec=kw,
                        num_defects=spec.num_defects,
                        defect_magnitude=spec.defect_magnitude,
                        params=jsonable(spec.params),
                        obf_params=jsonable(spec.obf_params),
                        tbf_params=jsonable(spec.tbf_params))
    with open(os.path.join(OUT, 'params.json'), 'w') as fp:
        json.dump(out, fp, indent=1, sort_keys=True)
    print('params.json', len(out))


def make_confs(spec, rng, num_random=4):
    """Fixed configurations: random, regular, near-contact, wrap-edge,
    cell-edge (positions on lattice-region boundaries)."""
    n = spec.boson_number
    L = spec.supercell_size
    confs = []
    for _ in range(num_random):
        c = spec.get_sys_conf_buffer()
        c[0] = L * rng.random_sample(n)
        confs.append(c)
    confs.append(spec.init_get_sys_conf(DIST_REGULAR))
    confs.append(spec.init_get_sys_conf(DIST_REGULAR, offset=0.3137))
    # near contact: pairs a few 1e-3..1e-9 apart, also across the boundary
    c = spec.get_sys_conf_buffer()
    c[0] = L * rng.random_sample(n)
    c[0, 1] = c[0, 0] + 1e-3
    c[0, 3] = (c[0, 2] - 1e-6) % L
    c[0, 4] = 1e-9
    c[0, 5] = L - 1e-9
    confs.append(c % L)
    # wrap edge: half the particles within 1e-2 of the box ends, exact 0,
    # exactly L/2 apart, exact cell boundaries
    c = spec.get_sys_conf_buffer()
    c[0] = L * rng.random_sample(n)
    c[0, 0] = 0.0
    c[0, 1] = 0.5 * L
    c[0, 2] = L - 1e-12
    c[0, 3] = 1.0 / (1 + spec.lattice_ratio)      # exactly z_a of cell 0
    c[0, 4] = 1.0
    c[0, 5] = 0.25 * L                              # rm from particle 0
    c[0, 6] = L - 0.25 * L
    for k in range(7, min(n, 12)):
        c[0, k] = (L - 1e-2 * rng.random_sample()) if k % 2 else \
            1e-2 * rng.random_sample()
    confs.append(c)
    return np.array(confs)


def gen_kernels():
    rng = np.random.RandomState(20261004)
    out = {}
    for tag, kw in SPECS.items():
        spec = mrbp_qmc.Spec(**kw)
        n = spec.boson_number
        cfc = spec.cfc_spec
        confs = make_confs(spec, rng, num_random=2 if n >= 512 else 4)
        nc = len(confs)
        wf = np.zeros(nc)
        en = np.zeros(nc)
        ith_e = np.zeros((nc, n))
        ith_f = np.zeros((nc, n))
        ith_e_only = np.zeros((nc, n))
        for k, c in enumerate(confs):
            wf[k] = core.wf_abs_log(c, *cfc)
            en[k] = core.energy(c, *cfc)
            d = core.drift(c, *cfc)
            for i in range(n):
                e_i, f_i = core.ith_energy_and_drift(i, c, *cfc)
                ith_e[k, i] = e_i
                ith_f[k, i] = f_i
                if n <= 128:
                    ith_e_only[k, i] = core.ith_energy(i, c, *cfc)
            assert np.array_equal(d[0], c[0])
            assert np.allclose(d[1], ith_f[k], rtol=0, atol=0)
            if n <= 128:
                assert np.array_equal(ith_e_only[k], ith_e[k])
        out[tag + '/pos'] = confs[:, 0, :].copy()
        out[tag + '/wf_abs_log'] = wf
        out[tag + '/energy'] = en
        out[tag + '/ith_energy'] = ith_e
        out[tag + '/ith_drift'] = ith_f
        print('kernels', tag, nc, wf[:2], en[:2])
    np.savez_compressed(os.path.join(OUT, 'kernels.npz'), **out)


def gen_vmc_tape():
    """RNG-tape VMC trajectories through `Sampling.blocks` (so the block
    bookkeeping of qmc_base/vmc.py:686-768 is exercised, including the
    energy carry on rejected moves across a block boundary)."""
    out = {}
    cases = [('box8', 0.125, 3, 96), ('box16', 0.125, 2, 128),
             ('free16', 0.125, 2, 64), ('deep16', 0.125, 2, 64),
             ('defect24', 0.3, 2, 48),
             # the benchmarked lane-group shape (one wavefront per walker) and
             # the position-classified pairs (r_m = L/2) on a two-particle shape
             ('box64', 0.125, 2, 24), ('odd24', 0.3, 2, 32)]
    for tag, spread, nblocks, ns in cases:
        spec = mrbp_qmc.Spec(**SPECS[tag])
        np.random.seed(4242)
        ini = spec.init_get_sys_conf()
        smp = mrbp_qmc.vmc.Sampling(spec, move_spread=spread, rng_seed=7)
        st0 = smp.build_state(ini)
        wf, en, ms, ar = [], [], [], []
        with harness.RNGTape() as tape:
            for blk in islice(smp.blocks(ns, st0), nblocks):
                wf.append(blk.iter_props.wf_abs_log.copy())
                en.append(blk.iter_props.energy.copy())
                ms.append(blk.iter_props.move_stat.copy())
                ar.append(blk.accept_rate)
                last = blk.last_state
        out[tag + '/ini_pos'] = ini[0].copy()
        out[tag + '/ini_wf_abs_log'] = np.float64(st0.wf_abs_log)
        out[tag + '/move_spread'] = np.float64(spread)
        out[tag + '/uniform'] = np.array(tape.uniform)
        out[tag + '/wf_abs_log'] = np.array(wf)
        out[tag + '/energy'] = np.array(en)
        out[tag + '/move_stat'] = np.array(ms)
        out[tag + '/accept_rate'] = np.array(ar)
        out[tag + '/last_pos'] = last.sys_conf[0].copy()
        assert not tape.normal
        print('vmc_tape', tag, len(tape.uniform), ar)
    np.savez_compressed(os.path.join(OUT, 'vmc_tape.npz'), **out)


def gen_dmc_tape():
    """RNG-tape DMC trajectories through `Sampling.states`: per time step the
    uniforms drawn by the branching loop and the normals drawn by the
    diffusion, plus every yielded scalar and the cloning table."""
    out = {}
    # (tag, spec, dt, target, max, kappa, steps, n_ini)
    cases = [('box8', 'box8', 1e-3, 24, 32, 0.5, 48, 24),
             ('box16', 'box16', 6.25e-4, 12, 16, 0.125, 24, 10),
             ('free16', 'free16', 1e-3, 12, 16, 0.5, 24, 16),
             # target above the cap: the population grows into max_num_walkers
             # and the branching loop truncates (qmc_base/dmc.py:638-651)
             ('cap8', 'box8', 2e-3, 40, 22, 0.5, 40, 20),
             ('box64', 'box64', 6.25e-4, 6, 8, 0.5, 10, 6),
             ('odd24', 'odd24', 1e-3, 8, 10, 0.5, 12, 8)]
    for tag, stag, dt, target, maxw, kappa, steps, n_ini in cases:
        spec = mrbp_qmc.Spec(**SPECS[stag])
        np.random.seed(1717)
        ini_set = np.array([spec.init_get_sys_conf() for _ in range(n_ini)])
        smp = mrbp_qmc.dmc.Sampling(spec, dt, max_num_walkers=maxw,
                                    target_num_walkers=target,
                                    num_walkers_control_factor=kappa,
                                    rng_seed=11, jit_parallel=False)
        st0 = smp.build_state(ini_set)
        rec = dict(energy=[], weight=[], num_walkers=[], ref_energy=[],
                   accum_energy=[], n_uniform=[], n_normal=[])
        refs = np.full((steps, maxw), -1, dtype=np.int64)
        conf_hist = np.zeros((steps, maxw, 2, spec.boson_number))
        en_hist = np.zeros((steps, maxw))
        with harness.RNGTape() as tape:
            nu = nn = 0
            for t, st in enumerate(islice(smp.states(st0), steps)):
                rec['energy'].append(st.energy)
                rec['weight'].append(st.weight)
                rec['num_walkers'].append(st.num_walkers)
                rec['ref_energy'].append(st.ref_energy)
                rec['accum_energy'].append(st.accum_energy)
                rec['n_uniform'].append(len(tape.uniform) - nu)
                rec['n_normal'].append(len(tape.normal) - nn)
                nu, nn = len(tape.uniform), len(tape.normal)
                nw = st.num_walkers
                refs[t, :nw] = st.branching_spec.cloning_ref[:nw]
                conf_hist[t, :nw] = st.confs[:nw]
                en_hist[t, :nw] = st.props.energy[:nw]
                assert np.all(st.props.weight[:nw] == 1.0)
                assert not st.props.mask[:nw].any()
                assert st.props.mask[nw:].all()
        out[tag + '/ini_pos'] = ini_set[:, 0, :].copy()
        out[tag + '/ini_energy'] = st0.props.energy[:n_ini].copy()
        out[tag + '/ini_drift'] = st0.confs[:n_ini, 1, :].copy()
        out[tag + '/ini_ref_energy'] = np.float64(st0.ref_energy)
        out[tag + '/cfg'] = np.array([dt, target, maxw, kappa, steps, n_ini])
        out[tag + '/uniform'] = np.array(tape.uniform)
        out[tag + '/normal'] = np.array(tape.normal)
        for k, v in rec.items():
            out[tag + '/' + k] = np.array(v)
        out[tag + '/cloning_ref'] = refs
        out[tag + '/confs'] = conf_hist
        out[tag + '/walker_energy'] = en_hist
        print('dmc_tape', tag, rec['num_walkers'][:8], max(rec['num_walkers']),
              rec['energy'][-1] / rec['weight'][-1])
    np.savez_compressed(os.path.join(OUT, 'dmc_tape.npz'), **out)


def gen_reblock():
    rng = np.random.RandomState(99)
    out = {}
    # AR(1) series: serially correlated like block totals of a Markov chain
    for tag, n, phi in [('ar1024', 1024, 0.6), ('ar512', 512, 0.0),
                        ('ar100', 100, 0.3), ('ar37', 37, 0.9)]:
        x = np.zeros(n)
        e = rng.normal(size=n)
        for k in range(1, n):
            x[k] = phi * x[k - 1] + e[k]
        x += 5.0
        w = 480.0 + 10 * rng.normal(size=n)
        otf = rb.on_the_fly_obj_create(x)
        obj = rb.OTFObject.from_non_obj_data(x)
        out[tag + '/x'] = x
        out[tag + '/w'] = w
        out[tag + '/otf_block_size'] = otf['BLOCK_SIZE'].copy()
        out[tag + '/otf_means_sum'] = otf['MEANS'].copy()
        out[tag + '/otf_means_sqr_sum'] = otf['MEANS_SQR'].copy()
        out[tag + '/otf_num_blocks'] = otf['NUM_BLOCKS'].copy()
        out[tag + '/block_sizes'] = np.array(obj.block_sizes)
        out[tag + '/num_blocks'] = np.array(obj.num_blocks)
        out[tag + '/means'] = np.array(obj.means)
        out[tag + '/vars'] = np.array(obj.vars)
        out[tag + '/errors'] = np.array(obj.errors)
        out[tag + '/iac_times'] = np.array(obj.iac_times)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            out[tag + '/opt_block_size'] = np.int64(obj.opt_block_size)
            out[tag + '/opt_iac_time'] = np.float64(obj.opt_iac_time)
            out[tag + '/eff_size'] = np.float64(obj.eff_size)
            out[tag + '/mean_eff_error'] = np.float64(obj.mean_eff_error)
            # qmc_exec/data/dmc.py:24-75 (ratio estimator) and vmc.py:23-42
            pb = dmc_data.PropBlocks(x * w, w)
            out[tag + '/dmc_mean'] = np.float64(pb.mean)
            out[tag + '/dmc_mean_error'] = np.float64(pb.mean_error)
            uw = dmc_data.UnWeightedPropBlocks(w)
            out[tag + '/uw_mean'] = np.float64(uw.mean)
            out[tag + '/uw_mean_error'] = np.float64(uw.mean_error)
            vb = vmc_data.PropBlocks(x)
            out[tag + '/vmc_mean'] = np.float64(vb.mean)
            out[tag + '/vmc_mean_error'] = np.float64(vb.mean_error)
        print('reblock', tag, out[tag + '/opt_block_size'],
              out[tag + '/dmc_mean'], out[tag + '/dmc_mean_error'])
    # 2-D (set) reblocking: the containers of S(k) / density block totals
    import warnings
    x2 = np.zeros((256, 5))
    e2 = rng.normal(size=(256, 5))
    for k in range(1, 256):
        x2[k] = 0.5 * x2[k - 1] + e2[k]
    x2 += np.arange(1, 6) * 3.0
    w2 = 480.0 + 10 * rng.normal(size=256)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        st = rb.OTFSet.from_non_obj_data(x2)
        out['set/x'] = x2
        out['set/w'] = w2
        out['set/block_sizes'] = np.array(st.block_sizes)
        out['set/num_blocks'] = np.array(st.num_blocks)
        out['set/means'] = np.array(st.means)
        out['set/vars'] = np.array(st.vars)
        out['set/iac_times'] = np.array(st.iac_times)
        out['set/opt_block_size'] = np.array(st.opt_block_size)
        out['set/opt_iac_time'] = np.array(st.opt_iac_time)
        out['set/eff_size'] = np.array(st.eff_size)
        out['set/mean_eff_error'] = np.array(st.mean_eff_error)
        pb = dmc_data.SSFPartBlocks(x2 * w2[:, None], w2[:, None])
        out['set/part_mean'] = np.array(pb.mean)
        out['set/part_mean_error'] = np.array(pb.mean_error)
        ssf3 = np.stack([x2 ** 2 + 40, x2, 0.1 * x2], axis=2) * w2[:, None, None]
        class _P:  # minimal PropsData stand-in (block weight totals)
            weight = w2
        sb = dmc_data.SSFBlocks.from_data(4, ssf3, _P, reduce_data=False,
                                          as_pure_est=False)
        out['set/ssf3'] = ssf3
        out['set/ssf_mean'] = np.array(sb.mean)
        out['set/ssf_mean_error'] = np.array(sb.mean_error)
    print('reblock set', out['set/opt_block_size'], out['set/part_mean'])
    np.savez_compressed(os.path.join(OUT, 'reblock.npz'), **out)


def gen_stats():
    """Seeds x runs table of block statistics on the N=16 box, produced by the
    reference with its own (numpy MT19937) streams: the 2-sigma gate."""
    out = {}
    spec = mrbp_qmc.Spec(**SPECS['box16'])
    # VMC: per seed 16 blocks x 512 steps, first 2 burned
    ns, nb, burn = 512, 16, 2
    vm = []
    for seed in range(1, 25):
        np.random.seed(1000 + seed)
        ini = spec.init_get_sys_conf()
        smp = mrbp_qmc.vmc.Sampling(spec, move_spread=0.125, rng_seed=seed)
        st0 = smp.build_state(ini)
        rows = []
        for b, blk in enumerate(islice(smp.blocks(ns, st0), nb)):
            if b < burn:
                continue
            e = blk.iter_props.energy
            rows.append((e.mean(), (e ** 2).mean(), blk.accept_rate))
        vm.append(rows)
        print('stats vmc seed', seed, np.mean([r[0] for r in rows]) / 16,
              np.mean([r[2] for r in rows]))
    out['vmc/block_stats'] = np.array(vm)      # [seed, block, (E, E2, acc)]
    out['vmc/cfg'] = np.array([ns, nb, burn, 0.125])
    # DMC: per seed target 96 / max 128, 10 blocks x 32 steps, first 2 burned
    dt, target, maxw, kappa, nts, nbd, burnd = 1e-3, 96, 128, 0.5, 32, 10, 2
    dm = []
    for seed in range(1, 11):
        np.random.seed(2000 + seed)
        ini = spec.init_get_sys_conf()
        vs = mrbp_qmc.vmc.Sampling(spec, move_spread=0.125, rng_seed=seed)
        chain = vs.as_chain(1024, vs.build_state(ini))
        ini_set = chain.confs[-target:]
        smp = mrbp_qmc.dmc.Sampling(spec, dt, max_num_walkers=maxw,
                                    target_num_walkers=target,
                                    num_walkers_control_factor=kappa,
                                    rng_seed=seed, jit_parallel=False)
        st0 = smp.build_state(ini_set)
        rows = []
        for b, blk in enumerate(islice(smp.blocks(st0, nts, burnd), nbd)):
            p = blk.iter_props
            rows.append((p.energy.sum(), p.weight.sum(),
                         float(p.num_walkers.sum())))
        dm.append(rows)
        r = np.array(rows[burnd:])
        print('stats dmc seed', seed, r[:, 0].sum() / r[:, 1].sum() / 16)
    out['dmc/block_totals'] = np.array(dm)     # [seed, block, (E, W, nw)]
    out['dmc/cfg'] = np.array([dt, target, maxw, kappa, nts, nbd, burnd])
    np.savez_compressed(os.path.join(OUT, 'stats.npz'), **out)


def gen_dmc_est():
    """DMC estimators: S(k) and density, mixed and pure
    (forward walking), through `Sampling.blocks` with a burned block so the
    gating and the per-block resets are exercised; RNG streams recorded so the
    runs can be replayed."""
    out = {}
    spec = mrbp_qmc.Spec(**SPECS['box8'])
    dt, target, maxw, kappa, nts, nblocks, burn = 1e-3, 20, 26, 0.5, 6, 3, 1
    cases = [('ssf_mixed', dict(ssf=(8, False, None))),
             ('ssf_pure', dict(ssf=(8, True, 4))),
             ('ssf_pure_full', dict(ssf=(5, True, None))),
             ('dens_mixed', dict(dens=(12, False, 99999999))),
             ('dens_pure', dict(dens=(12, True, 4))),
             ('both', dict(ssf=(8, True, None), dens=(16, True, 99999999)))]
    for tag, cfg in cases:
        np.random.seed(2718)
        ini_set = np.array([spec.init_get_sys_conf() for _ in range(18)])
        ssf_spec = dens_spec = None
        if 'ssf' in cfg:
            nm, pure, pfw = cfg['ssf']
            ssf_spec = mrbp_qmc.dmc.SSFEstSpec(nm, pure, pfw)
        if 'dens' in cfg:
            nb, pure, pfw = cfg['dens']
            dens_spec = mrbp_qmc.dmc.DensityEstSpec(nb, pure, pfw)
        smp = mrbp_qmc.dmc.Sampling(spec, dt, max_num_walkers=maxw,
                                    target_num_walkers=target,
                                    num_walkers_control_factor=kappa,
                                    rng_seed=3, density_est_spec=dens_spec,
                                    ssf_est_spec=ssf_spec, jit_parallel=False)
        st0 = smp.build_state(ini_set)
        ssf_blocks, dens_blocks, nws, nus, nns = [], [], [], [], []
        with harness.RNGTape() as tape:
            for blk in islice(smp.blocks(st0, nts, burn), nblocks):
                ssf_blocks.append(np.array(blk.iter_ssf, copy=True))
                dens_blocks.append(np.array(blk.iter_density, copy=True))
                nws.append(blk.iter_props.num_walkers.copy())
        out[tag + '/ini_pos'] = ini_set[:, 0, :].copy()
        out[tag + '/cfg'] = np.array([dt, target, maxw, kappa, nts, nblocks,
                                      burn])
        out[tag + '/uniform'] = np.array(tape.uniform)
        out[tag + '/normal'] = np.array(tape.normal)
        out[tag + '/num_walkers'] = np.array(nws).astype(np.int64)
        if ssf_spec is not None:
            out[tag + '/ssf_cfg'] = np.array([ssf_spec.num_modes,
                                              int(ssf_spec.as_pure_est),
                                              ssf_spec.pfw_num_time_steps])
            out[tag + '/iter_ssf'] = np.array(ssf_blocks)
        if dens_spec is not None:
            out[tag + '/dens_cfg'] = np.array([dens_spec.num_bins,
                                               int(dens_spec.as_pure_est),
                                               dens_spec.pfw_num_time_steps])
            out[tag + '/iter_density'] = np.array(dens_blocks)
        print('dmc_est', tag, np.array(nws)[:, -1],
              None if ssf_spec is None else np.array(ssf_blocks)[1, -1, 1],
              None if dens_spec is None else np.array(dens_blocks)[1, -1, :3, 0])
    np.savez_compressed(os.path.join(OUT, 'dmc_est.npz'), **out)


def gen_wf_opt():
    """Correlated-sampling variance of the local energy:
    the reference's CSWFOptimizer.principal_function on a fixed configuration
    set, for a scan of tbf_contact_cutoff values."""
    out = {}
    for tag in ('box16', 'deep16', 'free16'):
        spec = mrbp_qmc.Spec(**SPECS[tag])
        rng = np.random.RandomState(97)
        n, L = spec.boson_number, spec.supercell_size
        nconf = 96
        confs = np.zeros((nconf, 2, n))
        confs[:, 0, :] = L * rng.random_sample((nconf, n))
        # a few regular configurations with jitter
        reg = (np.arange(n) + 0.25) * L / n
        confs[:8, 0, :] = (reg + 0.05 * rng.standard_normal((8, n))) % L
        args = spec.cfc_spec
        ini_wf = np.array([core.wf_abs_log(c, args.model_params,
                                           args.obf_params, args.tbf_params)
                           for c in confs])
        opt = mrbp_qmc.CSWFOptimizer(spec, confs, ini_wf, num_workers=1)
        (lo, hi), = opt.principal_function_bounds
        cuts = np.linspace(lo, hi, 7)
        cuts[3] = spec.tbf_contact_cutoff
        var = np.array([opt.principal_function(c) for c in cuts])
        wf_en = [opt.wf_abs_log_and_energy_set(opt.update_spec(c).cfc_spec)
                 for c in cuts]
        out[tag + '/pos'] = confs[:, 0, :].copy()
        out[tag + '/ini_wf'] = ini_wf
        out[tag + '/bounds'] = np.array([lo, hi])
        out[tag + '/cutoffs'] = cuts
        out[tag + '/variance'] = var
        out[tag + '/wf_set'] = np.array([w for w, _ in wf_en])
        out[tag + '/energy_set'] = np.array([e for _, e in wf_en])
        print('wf_opt', tag, cuts, var)
    # weighed_variance on plain arrays
    rng = np.random.RandomState(5)
    wl, en = rng.standard_normal(50) * 3, 10 + rng.standard_normal(50)
    out['wv/weights_log'] = wl
    out['wv/energy'] = en
    out['wv/value'] = np.array(mrbp_qmc.CSWFOptimizer.weighed_variance(wl, en))
    np.savez_compressed(os.path.join(OUT, 'wf_opt.npz'), **out)


def gen_sweep():
    """Random model specs (depth, ratio, coupling, filling, cutoff, defects)
    with a few configurations each: parameter derivation and the kernel
    functions away from the r_m = L/4 unit-filling boxes."""
    rng = np.random.RandomState(20260)
    recs, k = [], 0
    out = {}
    while len(recs) < 36:
        n = int(rng.randint(3, 41))
        L = float(np.round(n * rng.uniform(0.6, 2.2), 3))
        kw = dict(lattice_depth=float(np.round(rng.choice(
                      [0.0, rng.uniform(0.5, 20), rng.uniform(20, 250)]), 4)),
                  lattice_ratio=float(np.round(rng.uniform(0.15, 4.0), 4)),
                  interaction_strength=float(np.round(
                      10 ** rng.uniform(-1.5, 2.0), 5)),
                  boson_number=n, supercell_size=L,
                  tbf_contact_cutoff=float(np.round(
                      L * rng.uniform(0.004, 0.4949), 5)))
        if rng.random_sample() < 0.3:
            cells = int(np.ceil(L))
            divs = [d for d in range(1, cells + 1) if cells % d == 0]
            kw['num_defects'] = int(rng.choice(divs))
            kw['defect_magnitude'] = float(np.round(rng.uniform(0, 60), 3))
        try:
            spec = mrbp_qmc.Spec(**kw)
            args = spec.cfc_spec
        except (ValueError, ZeroDivisionError, RuntimeError) as exc:
            print('sweep skip', kw, type(exc).__name__)
            continue
        confs = make_confs(spec, rng, num_random=3) if n >= 12 else \
            np.array([np.vstack([L * rng.random_sample(n), np.zeros(n)])
                      for _ in range(5)])
        wf = [core.wf_abs_log(c, *args) for c in confs]
        en = [core.energy(c, *args) for c in confs]
        ie = [[core.ith_energy_and_drift(i, c, *args) for i in range(n)]
              for c in confs]
        tag = 'sweep%02d' % len(recs)
        recs.append(dict(tag=tag, spec=kw, params=jsonable(spec.params),
                         obf_params=jsonable(spec.obf_params),
                         tbf_params=jsonable(spec.tbf_params)))
        out[tag + '/pos'] = confs[:, 0, :].copy()
        out[tag + '/wf_abs_log'] = np.array(wf)
        out[tag + '/energy'] = np.array(en)
        out[tag + '/ith'] = np.array(ie)      # [conf, i, (energy, drift)]
        print(tag, kw, wf[0], en[0])
    with open(os.path.join(OUT, 'sweep.json'), 'w') as fp:
        json.dump(recs, fp, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, 'sweep.npz'), **out)


def gen_vmc_extra():
    """What round 1 left unpinned on the VMC side:
      * the static structure factor of a single chain through
        `Sampling.blocks` (qmc_base/jastrow/vmc.py:304-351), including
        rejected steps -- where the reference copies the previous row, quirk
        D6 -- and a block boundary;
      * the Gaussian-proposal sampling `vmc_ndf` (qmc_base/vmc_ndf.py:43-59,
        mrbp_qmc/vmc_ndf.py:23-51): N normal draws then one uniform per step.
    """
    from phd_qmclib.mrbp_qmc import vmc_ndf
    out = {}
    spec = mrbp_qmc.Spec(**SPECS['box8'])
    n = spec.boson_number
    # -- S(k) of a single chain
    np.random.seed(515)
    ini = spec.init_get_sys_conf()
    smp = mrbp_qmc.vmc.Sampling(spec, move_spread=0.3, rng_seed=5,
                                ssf_est_spec=mrbp_qmc.vmc.SSFEstSpec(6))
    st0 = smp.build_state(ini)
    ssf, ms, wf, en = [], [], [], []
    with harness.RNGTape() as tape:
        for blk in islice(smp.blocks(40, st0), 2):
            ssf.append(np.array(blk.iter_ssf, copy=True))
            ms.append(blk.iter_props.move_stat.copy())
            wf.append(blk.iter_props.wf_abs_log.copy())
            en.append(blk.iter_props.energy.copy())
    assert not tape.normal
    out['ssf8/ini_pos'] = ini[0].copy()
    out['ssf8/move_spread'] = np.float64(0.3)
    out['ssf8/num_modes'] = np.int64(6)
    out['ssf8/uniform'] = np.array(tape.uniform)
    out['ssf8/iter_ssf'] = np.array(ssf)           # [block, step, mode, 3]
    out['ssf8/move_stat'] = np.array(ms)
    out['ssf8/wf_abs_log'] = np.array(wf)
    out['ssf8/energy'] = np.array(en)
    print('vmc_extra ssf8 rejected', int((~np.array(ms)).sum()), 'of',
          np.array(ms).size, np.array(ssf)[0, :3, 1])
    # -- Gaussian proposal
    np.random.seed(616)
    ini = spec.init_get_sys_conf()
    # (the attrs field order of the reference class starts with the inherited
    # `move_spread`, which the Gaussian sampling never reads)
    smp = vmc_ndf.Sampling(move_spread=0.0, model_spec=spec, time_step=0.01,
                           rng_seed=6)
    st0 = smp.build_state(ini)
    ms, wf, en, ar = [], [], [], []
    with harness.RNGTape() as tape:
        for blk in islice(smp.blocks(40, st0), 2):
            ms.append(blk.iter_props.move_stat.copy())
            wf.append(blk.iter_props.wf_abs_log.copy())
            en.append(blk.iter_props.energy.copy())
            ar.append(blk.accept_rate)
            last = blk.last_state
    # per step: N normals (proposal) then one uniform (Metropolis)
    kinds = ''.join(tape.kinds)
    nsteps = len(tape.uniform)
    assert kinds == ('n' * n + 'u') * nsteps, kinds[:40]
    rows = np.concatenate([np.array(tape.normal).reshape(nsteps, n),
                           np.array(tape.uniform)[:, None]], axis=1)
    out['ndf8/ini_pos'] = ini[0].copy()
    out['ndf8/time_step'] = np.float64(0.01)
    out['ndf8/tape'] = rows                        # [step, N + 1]
    out['ndf8/move_stat'] = np.array(ms)
    out['ndf8/wf_abs_log'] = np.array(wf)
    out['ndf8/energy'] = np.array(en)
    out['ndf8/accept_rate'] = np.array(ar)
    out['ndf8/last_pos'] = last.sys_conf[0].copy()
    print('vmc_extra ndf8', nsteps, ar)
    np.savez_compressed(os.path.join(OUT, 'vmc_extra.npz'), **out)


def gen_proc():
    """`Proc.exec` of the reference (qmc_exec/vmc/proc.py:87-250,
    qmc_exec/dmc/proc.py:136-415 through mrbp_qmc/{vmc,dmc}_exec/proc.py) on
    recorded RNG streams: the block totals / weight totals the driver hands
    to the reblocking containers, their means and errors, the default
    burn-in, with and without `keep_iter_data`."""
    import warnings
    from phd_qmclib.mrbp_qmc import dmc_exec, vmc_exec
    out = {}
    spec = mrbp_qmc.Spec(**SPECS['box8'])
    n = spec.boson_number

    def blocks_dump(prefix, blk):
        for name in ('totals', 'weight_totals'):
            v = getattr(blk, name, None)
            if v is not None:
                out[prefix + '/' + name] = np.array(v)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            out[prefix + '/mean'] = np.array(blk.mean)
            out[prefix + '/mean_error'] = np.array(blk.mean_error)

    # ---------------- VMC ----------------
    for tag, keep in (('vmc', False), ('vmc_keep', True)):
        np.random.seed(717)
        proc = vmc_exec.Proc(spec, move_spread=0.25, rng_seed=8, num_blocks=8,
                             num_steps_block=20, keep_iter_data=keep,
                             ssf_spec=vmc_exec.SSFEstSpec(num_modes=5))
        ini = spec.init_get_sys_conf()
        pin = vmc_exec.ProcInput(proc.sampling.build_state(ini))
        with harness.RNGTape() as tape:
            res = proc.exec(pin)
        assert not tape.normal
        out[tag + '/ini_pos'] = ini[0].copy()
        out[tag + '/cfg'] = np.array([0.25, 8, 20, int(keep), 5])
        out[tag + '/uniform'] = np.array(tape.uniform)
        blocks_dump(tag + '/energy', res.data.blocks.energy)
        blocks_dump(tag + '/ss_factor', res.data.blocks.ss_factor)
        out[tag + '/last_pos'] = res.state.sys_conf[0].copy()
        out[tag + '/last_wf_abs_log'] = np.float64(res.state.wf_abs_log)
        # default burn-in: num_blocks // 8 blocks (qmc_exec/vmc/proc.py:123-126)
        assert len(tape.uniform) == (8 + 1) * 20 * (n + 1) - (n + 1), \
            len(tape.uniform)
        print('proc', tag, out[tag + '/energy/totals'][:3],
              out[tag + '/energy/mean'])
    # ---------------- DMC ----------------
    for tag, keep, est in (('dmc', False, False), ('dmc_keep', True, False),
                           ('dmc_est', False, True)):
        np.random.seed(818)
        kw = {}
        if est:
            kw = dict(ssf_spec=dmc_exec.SSFEstSpec(num_modes=4,
                                                   as_pure_est=True),
                      density_spec=dmc_exec.DensityEstSpec(
                          num_bins=8, as_pure_est=False))
        proc = dmc_exec.Proc(spec, time_step=1e-3, max_num_walkers=48,
                             target_num_walkers=24, rng_seed=9, num_blocks=8,
                             num_time_steps_block=6, keep_iter_data=keep,
                             jit_parallel=False, **kw)
        # (the concrete class overrides the base post-init, so the default
        # stays None and `exec` burns num_blocks // 8 blocks)
        assert proc.burn_in_blocks is None
        assert proc.num_walkers_control_factor == 0.5
        ini_set = np.array([spec.init_get_sys_conf() for _ in range(20)])
        pin = dmc_exec.ProcInput(proc.sampling.build_state(ini_set))
        with harness.RNGTape() as tape:
            res = proc.exec(pin)
        # segment the streams per time step: uniforms (branching) then
        # normals (diffusion)
        kinds = ''.join(tape.kinds)
        n_u, n_n, i = [], [], 0
        while i < len(kinds):
            j = i
            while j < len(kinds) and kinds[j] == 'u':
                j += 1
            k = j
            while k < len(kinds) and kinds[k] == 'n':
                k += 1
            n_u.append(j - i)
            n_n.append(k - j)
            i = k
        assert len(n_u) == (8 + 1) * 6, len(n_u)
        out[tag + '/ini_pos'] = ini_set[:, 0, :].copy()
        out[tag + '/cfg'] = np.array([1e-3, 48, 24, 0.5, 8, 6, int(keep)])
        out[tag + '/uniform'] = np.array(tape.uniform)
        out[tag + '/normal'] = np.array(tape.normal)
        out[tag + '/n_uniform'] = np.array(n_u)
        out[tag + '/n_normal'] = np.array(n_n)
        b = res.data.blocks
        blocks_dump(tag + '/energy', b.energy)
        blocks_dump(tag + '/weight', b.weight)
        blocks_dump(tag + '/num_walkers', b.num_walkers)
        if est:
            blocks_dump(tag + '/ss_factor', b.ss_factor)
            blocks_dump(tag + '/density', b.density)
        out[tag + '/last_num_walkers'] = np.int64(res.state.num_walkers)
        out[tag + '/last_ref_energy'] = np.float64(res.state.ref_energy)
        print('proc', tag, out[tag + '/energy/totals'][:3],
              out[tag + '/energy/weight_totals'][:3], out[tag + '/energy/mean'],
              out[tag + '/num_walkers/totals'][:3])
    np.savez_compressed(os.path.join(OUT, 'proc_exec.npz'), **out)


ALL = dict(proc=gen_proc, vmc_extra=gen_vmc_extra, sweep=gen_sweep, wf_opt=gen_wf_opt, dmc_est=gen_dmc_est, params=gen_params, kernels=gen_kernels, vmc_tape=gen_vmc_tape,
           dmc_tape=gen_dmc_tape, reblock=gen_reblock, stats=gen_stats)

if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or list(ALL)
    for w in which:
        ALL[w]()
