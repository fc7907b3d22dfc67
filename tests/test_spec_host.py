"""Host-side model spec: derived parameters against the reference's values
(tests/golden/params.json) and the reference's validation errors
(mrbp_qmc/model.py:95-129, 174-196, 328-329)."""
from math import pi

import numpy as np
import pytest

from phd_qmclib_amd.mrbp_qmc import Spec, DIST_REGULAR, DIST_RAND

BASE = dict(lattice_depth=100, lattice_ratio=1, interaction_strength=1,
            boson_number=100, supercell_size=100, tbf_contact_cutoff=25)


def test_params_bit_identical_to_reference(golden_params):
    for tag, rec in golden_params.items():
        s = Spec(**rec['spec'])
        assert s.num_defects == rec['num_defects']
        assert s.defect_magnitude == rec['defect_magnitude']
        for name in ('params', 'obf_params', 'tbf_params'):
            mine = getattr(s, name)._asdict()
            for k, v in rec[name].items():
                assert mine[k] == v, (tag, name, k, mine[k], v)
        cfc = s.cfc_spec
        assert cfc.model_params == s.params


def test_frozen_and_extra_attribute():
    """tests/mrbp_qmc/test_model.py:32-42."""
    s = Spec(**BASE)
    with pytest.raises(AttributeError):
        setattr(s, 'extra_param', True)
    with pytest.raises(AttributeError):
        s.lattice_depth = 3.0


def test_validation_errors():
    with pytest.raises(ValueError, match='rm'):
        Spec(**dict(BASE, tbf_contact_cutoff=51))
    with pytest.raises(ValueError, match='negative'):
        Spec(**dict(BASE, num_defects=-1, defect_magnitude=1.0))
    with pytest.raises(ValueError, match='evenly'):
        Spec(**dict(BASE, num_defects=7, defect_magnitude=1.0))
    with pytest.raises(ValueError, match='greater'):
        Spec(**dict(BASE, num_defects=4, defect_magnitude=101.0))
    with pytest.raises(TypeError):
        Spec(**dict(BASE, boson_number=3.5))


def test_defect_defaults():
    """mrbp_qmc/model.py:174-196."""
    s = Spec(**BASE)
    assert s.num_defects == 0 and s.defect_magnitude == 100.0
    assert s.params.defects_sep == 1
    s = Spec(**dict(BASE, num_defects=4, defect_magnitude=50))
    assert s.params.defects_sep == 25 and s.defect_magnitude == 50.0
    s = Spec(**dict(BASE, num_defects=0, defect_magnitude=50))
    assert s.defect_magnitude == 100.0          # no defects: magnitude ignored
    s = Spec(**dict(BASE, defect_magnitude=50))
    assert s.num_defects == 0 and s.defect_magnitude == 100.0


def test_flags_and_geometry():
    s = Spec(**dict(BASE, lattice_depth=0))
    assert s.is_free and not s.is_ideal
    s = Spec(**dict(BASE, interaction_strength=0))
    assert s.is_ideal and s.tbf_params.param_k2 == 0.0 \
        and s.tbf_params.param_am == 1.0 and s.tbf_params.param_r_off == 50.0
    s = Spec(**dict(BASE, lattice_ratio=3))
    assert s.well_width == 0.25 and s.barrier_width == 0.75
    assert s.boundaries == (0., 100.)
    assert s.sys_conf_shape == (2, 100)


def test_init_get_sys_conf():
    """mrbp_qmc/model.py:248-273."""
    s = Spec(**BASE)
    reg = s.init_get_sys_conf(DIST_REGULAR)
    assert reg.shape == (2, 100)
    assert np.array_equal(reg[0], np.arange(100.0)) and not reg[1].any()
    off = s.init_get_sys_conf(DIST_REGULAR, offset=99.5)
    assert np.allclose(np.sort(off[0]), np.arange(100) + 0.5)
    np.random.seed(3)
    a = s.init_get_sys_conf(DIST_RAND)
    np.random.seed(3)
    b = s.init_get_sys_conf()
    assert np.array_equal(a, b) and np.all((a[0] >= 0) & (a[0] < 100))
    with pytest.raises(ValueError):
        s.init_get_sys_conf('nope')


def test_box_constants_of_survey():
    """SURVEY.md A.4 spot values (oracle session of the survey)."""
    s = Spec(lattice_depth=5 * pi ** 2, lattice_ratio=1,
             interaction_strength=2, boson_number=64, supercell_size=64,
             tbf_contact_cutoff=16)
    assert abs(s.obf_params.param_e0 - 14.453839759949091) < 1e-13
    t = s.tbf_params
    assert abs(t.param_k2 - 0.048378018700905141) < 1e-15
    assert abs(t.param_beta - 0.8306088619868438) < 1e-14
    assert abs(t.param_r_off - 30.475422080555774) < 1e-12
    assert abs(t.param_am - 0.98065445950531549) < 1e-14
