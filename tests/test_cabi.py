"""The C-ABI library loads and exports every symbol include/qmcwalk.h declares
(no compute calls: this runs without a GPU)."""
import os
import re

import pytest

from .conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'qmcwalk.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(qmc_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_expected_surface():
    syms = header_symbols()
    for must in ('qmc_engine_create', 'qmc_evaluate', 'qmc_vmc_run_block',
                 'qmc_dmc_run_block', 'qmc_dmc_step_local',
                 'qmc_dmc_export_walkers', 'qmc_last_error'):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from phd_qmclib_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    for name in header_symbols():
        assert hasattr(lib, name), name
    assert set(header_symbols()) == set(_lib.SIGNATURES)
    assert lib.qmc_abi_version() == 1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from phd_qmclib_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.QmcError, match='no CPU fallback'):
        _lib.load()


# ---- the model tables the kernels read, built and checked on the host ------
def _models():
    from math import pi
    from phd_qmclib_amd.mrbp_qmc import Spec
    out = []
    for n, cut in ((16, 0.25), (64, 0.25), (100, 0.1), (128, 0.25), (512, 0.05)):
        out.append(Spec(lattice_depth=5 * pi ** 2, lattice_ratio=1,
                        interaction_strength=2, boson_number=n,
                        supercell_size=n, tbf_contact_cutoff=cut * n))
    out.append(Spec(lattice_depth=20., lattice_ratio=0.5,
                    interaction_strength=10., boson_number=24,
                    supercell_size=24, tbf_contact_cutoff=3.0))
    return out


def test_trig_row_table_matches_long_double():
    """qmc_device.h trig_tab (row + angle addition) restated on the host by
    the library's diagnostic: at most an ulp from long-double sin / cos."""
    import ctypes as C
    from phd_qmclib_amd import _lib
    from phd_qmclib_amd.engine import model_params_struct
    lib = _lib.load()
    seen_rows = set()
    for spec in _models():
        mp = model_params_struct(spec.cfc_spec)
        rows, err = C.c_int32(0), C.c_double(0)
        assert lib.qmc_model_trig_table_info(C.byref(mp), C.byref(rows),
                                             C.byref(err)) == 0
        assert rows.value >= 256, 'every test model is tabulated'
        assert rows.value & (rows.value - 1) == 0
        assert err.value < 3.5e-16, (spec.boson_number, rows.value, err.value)
        seen_rows.add(rows.value)
    assert len(seen_rows) >= 1


def test_one_body_table_matches_closed_forms():
    import ctypes as C
    from phd_qmclib_amd import _lib
    from phd_qmclib_amd.engine import model_params_struct
    lib = _lib.load()
    for spec in _models():
        mp = model_params_struct(spec.cfc_spec)
        m1, m2, err = C.c_int32(0), C.c_int32(0), C.c_double(0)
        assert lib.qmc_model_one_body_table_info(
            C.byref(mp), C.byref(m1), C.byref(m2), C.byref(err)) == 0
        assert m1.value > 0 and m2.value > 0
        assert err.value < 5e-15


def test_log_row_table_matches_long_double():
    """qmc_math.h log_pos restated on the host by the library's diagnostic."""
    import ctypes as C
    from phd_qmclib_amd import _lib
    lib = _lib.load()
    rows, err = C.c_int32(0), C.c_double(0)
    assert lib.qmc_log_table_info(C.byref(rows), C.byref(err)) == 0
    assert rows.value == 256
    assert 0 < err.value < 2.5e-16
