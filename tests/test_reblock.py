"""Host-side statistics against the reference's outputs (tests/golden/
reblock.npz, produced by stats/reblock.py and qmc_exec/data/{vmc,dmc}.py) and
against the invariant the reference's own test asserts
(tests/stats/test_reblock.py:23-43: on-the-fly variances == direct ones)."""
import warnings

import numpy as np
import pytest

from phd_qmclib_amd.qmc_exec.data import dmc as dmc_data, vmc as vmc_data
from phd_qmclib_amd.stats import reblock as rb

TAGS = ['ar1024', 'ar512', 'ar100', 'ar37']


@pytest.mark.parametrize('tag', TAGS)
def test_otf_table_and_derived(golden_reblock, tag):
    g = golden_reblock
    x = g[tag + '/x']
    tab = rb.on_the_fly_obj_create(x)
    assert np.array_equal(tab['BLOCK_SIZE'], g[tag + '/otf_block_size'])
    assert np.array_equal(tab['NUM_BLOCKS'], g[tag + '/otf_num_blocks'])
    assert np.allclose(tab['MEANS'], g[tag + '/otf_means_sum'], rtol=1e-14)
    assert np.allclose(tab['MEANS_SQR'], g[tag + '/otf_means_sqr_sum'],
                       rtol=1e-14)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        obj = rb.OTFObject.from_non_obj_data(x)
        assert np.array_equal(obj.block_sizes, g[tag + '/block_sizes'])
        assert np.array_equal(obj.num_blocks, g[tag + '/num_blocks'])
        assert np.allclose(obj.means, g[tag + '/means'], rtol=1e-13)
        assert np.allclose(obj.vars, g[tag + '/vars'], rtol=1e-10)
        assert np.allclose(obj.errors, g[tag + '/errors'], rtol=1e-10)
        assert np.allclose(obj.iac_times, g[tag + '/iac_times'], rtol=1e-10)
        assert obj.opt_block_size == int(g[tag + '/opt_block_size'])
        assert np.isclose(obj.opt_iac_time, float(g[tag + '/opt_iac_time']),
                          rtol=1e-10)
        assert np.isclose(obj.eff_size, float(g[tag + '/eff_size']), rtol=1e-10)
        assert np.isclose(obj.mean_eff_error,
                          float(g[tag + '/mean_eff_error']), rtol=1e-10)


@pytest.mark.parametrize('tag', TAGS)
def test_block_containers(golden_reblock, tag):
    g = golden_reblock
    x, w = g[tag + '/x'], g[tag + '/w']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        pb = dmc_data.PropBlocks(x * w, w)
        assert np.isclose(pb.mean, float(g[tag + '/dmc_mean']), rtol=1e-12)
        assert np.isclose(pb.mean_error, float(g[tag + '/dmc_mean_error']),
                          rtol=1e-8)
        uw = dmc_data.UnWeightedPropBlocks(w)
        assert np.isclose(uw.mean, float(g[tag + '/uw_mean']), rtol=1e-13)
        assert np.isclose(uw.mean_error, float(g[tag + '/uw_mean_error']),
                          rtol=1e-10)
        vb = vmc_data.PropBlocks(x)
        assert np.isclose(vb.mean, float(g[tag + '/vmc_mean']), rtol=1e-13)
        assert np.isclose(vb.mean_error, float(g[tag + '/vmc_mean_error']),
                          rtol=1e-10)


def test_otf_equals_direct_reblocking():
    """The reference's own invariant: variances of the hierarchical block
    means equal those of directly reshaped blocks."""
    rng = np.random.RandomState(0)
    x = rng.random_sample(2 ** 12)
    obj = rb.OTFObject.from_non_obj_data(x, min_num_blocks=32)
    for B, v in zip(obj.block_sizes, obj.vars):
        nb = len(x) // B
        direct = x[:nb * B].reshape(nb, B).mean(axis=1).var(ddof=1)
        assert np.isclose(v, direct, rtol=1e-10)


def test_opt_block_size_warning_and_fallback():
    """tests/stats/test_reblock.py:46-65."""
    x = np.random.RandomState(1).random_sample(2)
    with pytest.warns(RuntimeWarning):
        obj = rb.OTFObject.from_non_obj_data(x)
        assert obj.opt_block_size == obj.block_sizes.max()


def test_concat_blocks():
    """tests/qmc_exec/test_data_dmc.py: block containers concatenate."""
    a = dmc_data.EnergyBlocks(np.arange(4.), np.ones(4))
    b = dmc_data.EnergyBlocks(np.arange(6.), np.ones(6))
    c = a + b
    assert len(c) == 10 and isinstance(c, dmc_data.EnergyBlocks)
    assert len(dmc_data.WeightBlocks(np.ones(3)) +
               dmc_data.WeightBlocks(np.ones(5))) == 8


def test_set_reblocking_and_estimator_containers(golden_reblock):
    """2-D reblocking (OTFSet) and the S(k) block containers against the
    reference (stats/reblock.py:759-923, qmc_exec/data/dmc.py:396-621)."""
    g = golden_reblock
    x2, w2 = g['set/x'], g['set/w']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        st = rb.OTFSet.from_non_obj_data(x2)
        assert np.array_equal(st.block_sizes, g['set/block_sizes'])
        assert np.array_equal(st.num_blocks, g['set/num_blocks'])
        assert np.allclose(st.means, g['set/means'], rtol=1e-13)
        assert np.allclose(st.vars, g['set/vars'], rtol=1e-9)
        assert np.allclose(st.iac_times, g['set/iac_times'], rtol=1e-9)
        assert np.array_equal(st.opt_block_size, g['set/opt_block_size'])
        assert np.allclose(st.opt_iac_time, g['set/opt_iac_time'], rtol=1e-9)
        assert np.allclose(st.eff_size, g['set/eff_size'], rtol=1e-9)
        assert np.allclose(st.mean_eff_error, g['set/mean_eff_error'],
                           rtol=1e-9)
        assert np.isclose(st[2].mean, g['set/means'][2, 0], rtol=1e-13)
        pb = dmc_data.SSFPartBlocks(x2 * w2[:, None], w2[:, None])
        assert np.allclose(pb.mean, g['set/part_mean'], rtol=1e-12)
        assert np.allclose(pb.mean_error, g['set/part_mean_error'], rtol=1e-7)

        class P:
            weight = w2
        sb = dmc_data.SSFBlocks.from_data(4, g['set/ssf3'], P,
                                          reduce_data=False, as_pure_est=False)
        assert np.allclose(sb.mean, g['set/ssf_mean'], rtol=1e-10)
        assert np.allclose(sb.mean_error, g['set/ssf_mean_error'], rtol=1e-7)
