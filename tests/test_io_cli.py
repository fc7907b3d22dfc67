"""Result files and the command-line front end:
HDF5 layout of the reference (qmc_exec/io.py, qmc_exec/{vmc,dmc}/io.py,
qmc_exec/data/*.py hdf5_export) through h5py or the libhdf5 facade, the
configuration loader and `CLIApp`."""
import os
import subprocess
import textwrap

import numpy as np
import pytest

from phd_qmclib_amd.util import h5lite

try:
    h5lite.open_file  # noqa: B018
    h5lite._load()
    HAVE_HDF5 = True
except h5lite.HDF5Unavailable:          # pragma: no cover
    try:
        import h5py  # noqa: F401
        HAVE_HDF5 = True
    except ImportError:
        HAVE_HDF5 = False

needs_hdf5 = pytest.mark.skipif(not HAVE_HDF5, reason='no HDF5 library')
CONDA_PY = '/opt/conda/bin/python3.9'

MODEL = dict(lattice_depth=24, lattice_ratio=1, interaction_strength=1.0,
             boson_number=16, supercell_size=16.0, tbf_contact_cutoff=4)


def _yaml_config(tmp_path, kind, extra_proc, out='out/res.h5', second=None):
    meta = textwrap.dedent('''\
        meta:
          name: "test run"
          description: "d"
          author: "a"
          institution: "i"
          author_email: "e@x"
          category: "c"
          tags: ["qmc", "%s"]
        ''' % kind)
    proc = textwrap.dedent('''\
        app_spec:
          - proc:
              model_spec:
                lattice_depth: 24
                lattice_ratio: 1
                interaction_strength: 1.0
                boson_number: 16
                supercell_size: 16.0
                tbf_contact_cutoff: 4
        %s
            proc_input:
              type: "MODEL_SYS_CONF"
              dist_type: "RANDOM"
            proc_output:
              type: "HDF5_FILE"
              location: "%s"
              group: "%s-proc-ID0"
            proc_id: 101
        ''') % (textwrap.indent(textwrap.dedent(extra_proc), ' ' * 6), out, kind)
    text = meta + proc + textwrap.indent(second or '', '  ')
    path = tmp_path / f'{kind}-spec.yml'
    path.write_text(text)
    return path


SECOND_DMC_PROC = textwrap.dedent('''\
    - proc:
        model_spec:
          lattice_depth: 24
          lattice_ratio: 1
          interaction_strength: 1.0
          boson_number: 16
          supercell_size: 16.0
          tbf_contact_cutoff: 4
        time_step: 1e-3
        rng_seed: 8
        num_blocks: 4
        num_time_steps_block: 16
        burn_in_blocks: 0
        max_num_walkers: 64
        target_num_walkers: 48
      proc_input:
        type: "HDF5_FILE"
        location: "dmc.h5"
        group: "dmc-proc-ID0"
      proc_output:
        type: "HDF5_FILE"
        location: "dmc.h5"
        group: "dmc-proc-ID1"
    ''')


@needs_hdf5
def test_h5lite_types_roundtrip(tmp_path):
    p = tmp_path / 't.h5'
    rec = np.zeros(3, dtype=[('CLONING_FACTOR', np.int32),
                             ('CLONING_REF', np.int32)])
    rec['CLONING_REF'] = [2, 0, 1]
    with h5lite.File(p, 'w') as f:
        g = f.require_group('a/b/dmc')
        g.attrs.update({'x': 1.5, 'n': 7, 'flag': True, 'name': 'héllo',
                        'arr': np.arange(4.0)})
        g.create_dataset('e', data=np.linspace(0, 1, 5))
        g.create_dataset('i', data=np.arange(6, dtype=np.int64).reshape(2, 3))
        g.create_dataset('u', data=np.arange(3, dtype=np.uint64))
        g.create_dataset('m', data=np.array([True, False, True]))
        g.create_dataset('s', data=3.25)
        g.create_dataset('rec', data=rec)
        with pytest.raises(TypeError):
            g.attrs['none'] = None
        with pytest.raises(ValueError):
            g.create_dataset('e', data=np.zeros(2))
    with h5lite.File(p, 'r') as f:
        assert 'a/b/dmc' in f and 'a/x' not in f and f.get('nope') is None
        g = f['a/b/dmc']
        assert sorted(g.keys()) == ['e', 'i', 'm', 'rec', 's', 'u']
        at = dict(g.attrs.items())
        assert at['x'] == 1.5 and at['n'] == 7 and at['flag'] == True  # noqa
        assert at['name'] == 'héllo' and np.array_equal(at['arr'], np.arange(4.0))
        assert np.array_equal(g['e'][()], np.linspace(0, 1, 5))
        assert g['i'][()].dtype == np.int64 and g['i'].shape == (2, 3)
        assert g['u'][()].dtype == np.uint64
        assert g['m'][()].dtype == bool and g['m'][()].tolist() == [True, False, True]
        assert g['s'][()] == 3.25
        assert np.array_equal(g['rec'][()], rec)
        with pytest.raises(KeyError):
            f['zz']
    with h5lite.File(p, 'a') as f:
        del f['a/b']['dmc']
        assert f['a/b'].keys() == []


@needs_hdf5
@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason='no second interpreter')
def test_h5lite_files_are_h5py_files(tmp_path):
    """Files written by the facade are read by the real h5py with the same
    types, and the other way round."""
    ours, theirs = tmp_path / 'ours.h5', tmp_path / 'theirs.h5'
    with h5lite.File(ours, 'w') as f:
        g = f.require_group('g/dmc/state')
        g.attrs.update(energy=-1.25, num_walkers=480, done=False, tag='abc')
        g.create_dataset('confs', data=np.arange(24.).reshape(2, 2, 6))
        g.create_dataset('mask', data=np.array([True, False]))
    script = textwrap.dedent(f'''
        import h5py, numpy as np
        f = h5py.File({str(ours)!r}, 'r'); g = f['g/dmc/state']
        a = dict(g.attrs.items())
        assert a['energy'] == -1.25 and a['num_walkers'] == 480
        assert a['done'] is np.False_ or a['done'] == False
        assert isinstance(a['tag'], str) and a['tag'] == 'abc'
        assert g['confs'].shape == (2, 2, 6) and g['confs'].dtype == np.float64
        assert g['mask'].dtype == np.bool_ and g['mask'][()].tolist() == [True, False]
        f.close()
        f = h5py.File({str(theirs)!r}, 'w'); g = f.require_group('q/vmc')
        g.attrs.update(dict(a=2.5, b=3, c=True, d='text'))
        g.create_dataset('v', data=np.arange(5.))
        g.create_dataset('mask', data=np.array([True, True, False]))
        f.close()
        print('OK')
        ''')
    try:
        out = subprocess.run([CONDA_PY, '-c', script], capture_output=True,
                             text=True, timeout=120)
    except (OSError, subprocess.TimeoutExpired):
        pytest.skip('second interpreter not runnable')
    if 'No module named' in out.stderr:
        pytest.skip('h5py missing in the second interpreter')
    assert out.returncode == 0 and 'OK' in out.stdout, out.stderr[-2000:]
    with h5lite.File(theirs, 'r') as f:
        g = f['q/vmc']
        a = dict(g.attrs.items())
        assert a['a'] == 2.5 and a['b'] == 3 and a['c'] == True and a['d'] == 'text'  # noqa
        assert np.array_equal(g['v'][()], np.arange(5.))
        assert g['mask'][()].tolist() == [True, True, False]


@needs_hdf5
def test_dmc_result_file_roundtrip(tmp_path):
    from phd_qmclib_amd.mrbp_qmc import dmc_exec
    from phd_qmclib_amd.qmc_base import dmc as dmc_base
    from phd_qmclib_amd.qmc_exec.data import dmc as dd
    from phd_qmclib_amd.qmc_exec.io import HDF5FileHandlerGroupError
    proc = dmc_exec.Proc.from_config(dict(
        model_spec=MODEL, time_step=1e-3, num_blocks=6, num_time_steps_block=8,
        max_num_walkers=12, target_num_walkers=10, rng_seed=5,
        ssf_spec=dict(num_modes=4, as_pure_est=True),
        density_spec=dict(num_bins=5)))
    rng = np.random.RandomState(0)
    props = dmc_base.StateProps(rng.rand(12), rng.rand(12),
                                np.arange(12) >= 10)
    state = dmc_base.State(
        confs=rng.rand(12, 2, 16), props=props, energy=3.5, weight=9.75,
        num_walkers=10, ref_energy=0.35, accum_energy=0.36, max_num_walkers=12,
        branching_spec=dmc_base.BranchingSpec(np.ones(12, np.int64),
                                              np.arange(12)[::-1].copy()))
    w = rng.rand(6)
    part = lambda: dd.SSFPartBlocks(rng.rand(6, 4), np.tile(w[:, None], (1, 4)))  # noqa
    data = dd.SamplingData(dd.PropsDataBlocks(
        dd.EnergyBlocks(rng.rand(6), w), dd.WeightBlocks(w),
        dd.NumWalkersBlocks(rng.randint(8, 12, 6).astype(np.uint64)),
        dd.DensityBlocks(rng.rand(6, 5), np.tile(w[:, None], (1, 5))),
        dd.SSFBlocks(part(), part(), part())))
    res = dmc_exec.ProcResult(state, proc, data)
    h = dmc_exec.HDF5FileHandler(str(tmp_path / 'sub' / 'r.h5'), 'run-A')
    assert h.type == 'HDF5_FILE' and h.sampling_type == 'dmc'
    h.dump(res)
    with pytest.raises(HDF5FileHandlerGroupError):
        h.dump(res)
    dmc_exec.HDF5FileHandler(h.location, 'run-A', dump_replace=True).dump(res)
    # the reference's tree
    with h5lite.open_file(h.location, 'r') as f:
        q = f['run-A/dmc']
        assert sorted(q.keys()) == ['data', 'proc_spec', 'state']
        assert sorted(q['state'].keys()) == ['branching_spec', 'confs', 'props']
        assert sorted(q['proc_spec'].keys()) == ['density_spec', 'model_spec',
                                                 'ssf_spec']
        assert sorted(q['data/blocks'].keys()) == [
            'density', 'energy', 'num_walkers', 'ss_factor', 'weight']
        assert sorted(q['data/blocks/ss_factor'].keys()) == [
            'fdk_imag', 'fdk_real', 'fdk_sqr_abs']
        assert sorted(q['data/blocks/energy'].keys()) == ['totals',
                                                          'weight_totals']
        assert q['proc_spec'].attrs['time_step'] == 1e-3
        assert q['proc_spec/model_spec'].attrs['boson_number'] == 16
    back = h.load()
    assert back.proc == proc
    assert np.array_equal(back.state.confs, state.confs)
    assert np.array_equal(back.state.props.mask, props.mask)
    assert back.state.props.mask.dtype == bool
    assert np.array_equal(back.state.branching_spec.cloning_ref,
                          state.branching_spec.cloning_ref)
    for f_ in ('energy', 'weight', 'num_walkers', 'ref_energy', 'accum_energy',
               'max_num_walkers'):
        assert getattr(back.state, f_) == getattr(state, f_)
    b0, b1 = data.blocks, back.data.blocks
    assert np.array_equal(b1.energy.totals, b0.energy.totals)
    assert np.array_equal(b1.energy.weight_totals, b0.energy.weight_totals)
    assert np.array_equal(b1.num_walkers.totals, b0.num_walkers.totals)
    assert np.array_equal(b1.density.totals, b0.density.totals)
    assert np.array_equal(b1.ss_factor.fdk_imag_part.totals,
                          b0.ss_factor.fdk_imag_part.totals)
    assert b1.energy.mean == b0.energy.mean
    with pytest.raises(KeyError):
        dmc_exec.HDF5FileHandler(h.location, 'other').load()


@needs_hdf5
def test_vmc_result_file_roundtrip(tmp_path):
    from phd_qmclib_amd.mrbp_qmc import vmc_exec
    from phd_qmclib_amd.qmc_base import vmc as vmc_base
    from phd_qmclib_amd.qmc_exec.data import vmc as vd
    proc = vmc_exec.Proc.from_config(dict(
        model_spec=MODEL, move_spread=0.125, num_blocks=5, num_steps_block=16,
        rng_seed=2, ssf_spec=dict(num_modes=3)))
    rng = np.random.RandomState(1)
    state = vmc_base.State(rng.rand(2, 16), -3.25, 1)
    data = vd.SamplingData(vd.PropsDataBlocks(
        vd.EnergyBlocks(rng.rand(5)),
        vd.SSFBlocks.from_data(rng.rand(5, 3, 3), reduce_data=False)))
    h = vmc_exec.HDF5FileHandler(tmp_path / 'v.h5', 'g0')
    h.dump(vmc_exec.ProcResult(state, proc, data))
    back = h.load()
    assert back.proc == proc
    assert np.array_equal(back.state.sys_conf, state.sys_conf)
    assert back.state.wf_abs_log == -3.25 and back.state.move_stat == 1
    assert np.array_equal(back.data.blocks.energy.totals,
                          data.blocks.energy.totals)
    for part in ('fdk_sqr_abs_part', 'fdk_real_part', 'fdk_imag_part'):
        assert np.array_equal(getattr(back.data.blocks.ss_factor, part).totals,
                              getattr(data.blocks.ss_factor, part).totals)
    assert back.data.blocks.ss_factor.mean.shape == (3,)
    with h5lite.open_file(h.location, 'r') as f:
        assert sorted(f['g0/vmc/data/blocks/ss_factor'].keys()) == [
            'fdk_imag', 'fdk_real', 'fdk_sqr_abs']
        assert sorted(f['g0/vmc/state'].attrs.keys()) == ['move_stat',
                                                          'wf_abs_log']


def test_config_loader_and_app(tmp_path):
    from phd_qmclib_amd.mrbp_qmc import dmc_cli, dmc_exec, vmc_exec
    path = _yaml_config(tmp_path, 'dmc', '''\
        time_step: 1e-3
        num_batches: 4
        num_time_steps_batch: 8
        burn_in_batches: null
        max_num_walkers: 64
        target_num_walkers: 48
        ssf_spec:
          num_modes: 6
          as_pure_est: true
        ''')
    cfg = dmc_exec.config_loader.load(path)
    out = cfg['app_spec'][0]['proc_output']
    assert out['location'] == str(tmp_path / 'out/res.h5')   # made absolute
    with pytest.warns(DeprecationWarning):
        app = dmc_exec.CLIApp.from_config(cfg)
    spec = app.app_spec[0]
    assert spec.proc_id == 101 and spec.proc.num_blocks == 4
    assert spec.proc.num_time_steps_block == 8 and spec.proc.burn_in_blocks is None
    assert spec.proc.ssf_spec.num_modes == 6 and spec.proc.time_step == 1e-3
    assert isinstance(spec.proc_input, dmc_exec.ModelSysConfSpec)
    assert isinstance(spec.proc_output, dmc_exec.HDF5FileHandler)
    assert app.meta.tags == '#qmc - #dmc'
    # deprecated aliases and error paths
    text = path.read_text().replace('proc_input:', 'input:').replace(
        'proc_output:', 'output:')
    alias = tmp_path / 'alias.yaml'
    alias.write_text(text)
    cfg2 = dmc_exec.config_loader.load(alias)
    assert 'proc_input' in cfg2['app_spec'][0]
    bad = tmp_path / 'conf.txt'
    bad.write_text(text)
    with pytest.raises(IOError):
        dmc_exec.config_loader.load(bad)
    with pytest.raises(IOError):
        dmc_exec.config_loader.load(tmp_path / 'noext')
    with pytest.raises(TypeError):
        dmc_exec.get_io_handler({'type': 'CSV_FILE'})
    with pytest.raises(TypeError):
        vmc_exec.AppSpec.from_config(dict(
            proc=dict(model_spec=MODEL, move_spread=0.1),
            proc_input=dict(type='MODEL_SYS_CONF', dist_type='RANDOM'),
            proc_output=dict(type='MODEL_SYS_CONF', dist_type='RANDOM')))
    # a second procedure that continues from the first one's result file
    (tmp_path / 'two').mkdir()
    two = _yaml_config(tmp_path / 'two', 'dmc', 'time_step: 1e-3\n',
                       out='two.h5', second=SECOND_DMC_PROC)
    app2 = dmc_exec.CLIApp.from_config(dmc_exec.config_loader.load(two))
    assert len(app2.app_spec) == 2 and app2.app_spec[1].proc_id == 1
    assert isinstance(app2.app_spec[1].proc_input, dmc_exec.HDF5FileHandler)
    assert app2.app_spec[1].proc_input.group == 'dmc-proc-ID0'
    # dry run through the command line: validates, executes nothing
    with pytest.warns(DeprecationWarning):
        assert dmc_cli.main(['start', str(path), '--dry-run', '-S']) == 0
    assert not (tmp_path / 'out/res.h5').exists()


@pytest.mark.gpu
@needs_hdf5
def test_cli_end_to_end(tmp_path):
    """`mrbp-vmc start` then `mrbp-dmc start` with two procedures, the second
    continuing from the first one's result file."""
    from phd_qmclib_amd.mrbp_qmc import dmc_cli, dmc_exec, vmc_cli, vmc_exec
    vpath = _yaml_config(tmp_path, 'vmc', '''\
        move_spread: 0.125
        rng_seed: 4
        num_blocks: 6
        num_steps_block: 64
        ssf_spec:
          num_modes: 5
        ''', out='vmc.h5')
    app, results = vmc_cli.start(vpath, silent=True)
    vres = vmc_exec.HDF5FileHandler(tmp_path / 'vmc.h5', 'vmc-proc-ID0').load()
    assert vres.proc == app.app_spec[0].proc
    assert np.array_equal(vres.data.blocks.energy.totals,
                          results[0].data.blocks.energy.totals)
    assert vres.data.blocks.ss_factor.fdk_real_part.totals.shape == (6, 5)
    assert np.array_equal(vres.state.sys_conf, results[0].state.sys_conf)
    dpath = _yaml_config(tmp_path, 'dmc', '''\
        time_step: 1e-3
        rng_seed: 7
        num_blocks: 4
        num_time_steps_block: 16
        burn_in_blocks: 1
        max_num_walkers: 64
        target_num_walkers: 48
        ssf_spec:
          num_modes: 6
          as_pure_est: true
        ''', out='dmc.h5', second=SECOND_DMC_PROC)
    app, results = dmc_cli.start(dpath, silent=True)
    assert len(results) == 2
    r0 = dmc_exec.HDF5FileHandler(tmp_path / 'dmc.h5', 'dmc-proc-ID0').load()
    r1 = dmc_exec.HDF5FileHandler(tmp_path / 'dmc.h5', 'dmc-proc-ID1').load()
    assert r0.proc == app.app_spec[0].proc and r1.proc == app.app_spec[1].proc
    assert np.array_equal(r0.data.blocks.energy.totals,
                          results[0].data.blocks.energy.totals)
    assert r0.data.blocks.ss_factor.fdk_real_part.totals.shape == (4, 6)
    assert r1.data.blocks.ss_factor is None
    assert np.array_equal(r1.state.confs, results[1].state.confs)
    # the second procedure started from the first one's last state
    assert r0.state.num_walkers == results[0].state.num_walkers
    e = r1.data.blocks.energy
    assert np.isfinite(e.mean) and 5 < e.mean / 16 < 30
