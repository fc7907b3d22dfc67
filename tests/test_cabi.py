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
