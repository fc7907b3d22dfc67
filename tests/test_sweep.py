"""Random model specs (tests/golden/sweep.{json,npz}, produced by the
reference through oracle/refgen/gen_golden.py sweep): lattice depth 0..250,
ratio 0.15..4, coupling 0.03..100, filling 0.45..1.7, contact cutoff
0.004 L..0.495 L, lattice defects.  Host parameter derivation, C oracle and
the device kernels away from the r_m = L/4 unit-filling boxes."""
import json
import os

import numpy as np
import pytest

from .conftest import GOLDEN

RTOL = 2e-11


def _records():
    with open(os.path.join(GOLDEN, 'sweep.json')) as fp:
        return json.load(fp)


RECS = _records()
TAGS = [r['tag'] for r in RECS]


@pytest.fixture(scope='module')
def sweep():
    return np.load(os.path.join(GOLDEN, 'sweep.npz'))


@pytest.mark.parametrize('rec', RECS, ids=TAGS)
def test_host_params_match_reference(rec):
    from phd_qmclib_amd.mrbp_qmc import Spec
    s = Spec(**rec['spec'])
    for name in ('params', 'obf_params', 'tbf_params'):
        mine = getattr(s, name)._asdict()
        for k, v in rec[name].items():
            # root finders (brentq / findroot) of another scipy / mpmath
            # version may stop one iteration apart: 1e-13, exact in practice
            if isinstance(v, float):
                assert mine[k] == pytest.approx(v, rel=1e-13, abs=1e-300), \
                    (rec['tag'], name, k, mine[k], v)
            else:
                assert mine[k] == v, (rec['tag'], name, k)


@pytest.mark.parametrize('rec', RECS, ids=TAGS)
def test_oracle_bit_exact(oracle, sweep, rec):
    tag = rec['tag']
    m = oracle.model_from_params(rec['params'], rec['obf_params'],
                                 rec['tbf_params'])
    wf, en, ie, fd = oracle.evaluate_set(m, sweep[tag + '/pos'])
    assert np.array_equal(wf, sweep[tag + '/wf_abs_log'])
    assert np.array_equal(en, sweep[tag + '/energy'])
    assert np.array_equal(ie, sweep[tag + '/ith'][:, :, 0])
    assert np.array_equal(fd, sweep[tag + '/ith'][:, :, 1])


@pytest.mark.gpu
@pytest.mark.parametrize('rec', RECS, ids=TAGS)
def test_device_vs_reference(sweep, rec):
    from phd_qmclib_amd.engine import ModelEngine
    from phd_qmclib_amd.mrbp_qmc import Spec
    tag = rec['tag']
    eng = ModelEngine(Spec(**rec['spec']).cfc_spec)
    out = eng.evaluate(sweep[tag + '/pos'])
    eng.close()
    ref_ie = sweep[tag + '/ith'][:, :, 0]
    ref_fd = sweep[tag + '/ith'][:, :, 1]
    for name, got, ref in [('wf', out.wf_abs_log, sweep[tag + '/wf_abs_log']),
                           ('energy', out.energy, sweep[tag + '/energy']),
                           ('ith', out.ith_energy, ref_ie),
                           ('drift', out.drift, ref_fd)]:
        # near-contact configurations put 1e9-sized terms into a particle's
        # energy and drift: scale by the largest magnitude of the quantity
        # within the configuration (the walker sums see the same terms)
        scale = np.maximum(1.0, np.abs(ref).reshape(ref.shape[0], -1).max(1))
        if name in ('wf', 'energy'):
            scale = np.maximum(scale, np.abs(ref_ie).max(1))
        err = np.abs(got - ref).reshape(ref.shape[0], -1).max(1) / scale
        assert err.max() <= RTOL, (tag, name, err)
