"""ctypes binding of libqmcwalk.so (C-ABI: include/qmcwalk.h).

The HIP library is the only compute path of this package.  If it has not been
built (`python -c "import __graft_entry__ as g; g.build()"` or
`make -C phd_qmclib_amd/csrc`) loading fails with an explicit error -- there
is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (QMCWALK_LIB: development knob to A/B another build of the same library)
LIB_PATH = os.environ.get('QMCWALK_LIB') or os.path.join(_HERE, 'libqmcwalk.so')

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_i64p = C.POINTER(C.c_int64)
_u64p = C.POINTER(C.c_uint64)
_vp = C.c_void_p


class QmcError(RuntimeError):
    """An error reported by libqmcwalk.so."""


class ModelParams(C.Structure):
    """qmc_model_params (include/qmcwalk.h)."""
    _fields_ = [
        ('lattice_depth', C.c_double), ('lattice_ratio', C.c_double),
        ('interaction_strength', C.c_double), ('boson_number', C.c_int64),
        ('supercell_size', C.c_double), ('tbf_contact_cutoff', C.c_double),
        ('defect_magnitude', C.c_double), ('defects_sep', C.c_int64),
        ('well_width', C.c_double), ('barrier_width', C.c_double),
        ('is_free', C.c_int64), ('is_ideal', C.c_int64),
        ('param_e0', C.c_double), ('param_k1', C.c_double),
        ('param_kp1', C.c_double), ('param_k2', C.c_double),
        ('param_beta', C.c_double), ('param_r_off', C.c_double),
        ('param_am', C.c_double),
    ]


class VmcParams(C.Structure):
    _fields_ = [('num_chains', C.c_int64), ('move_spread', C.c_double),
                ('rng_seed', C.c_uint64), ('chain0', C.c_uint32),
                ('gaussian', C.c_int32)]


class DmcParams(C.Structure):
    _fields_ = [('max_num_walkers', C.c_int64),
                ('target_num_walkers', C.c_int64),
                ('time_step', C.c_double),
                ('num_walkers_control_factor', C.c_double),
                ('rng_seed', C.c_uint64), ('slot0', C.c_uint32),
                ('fix_stale_energy', C.c_int32),
                ('external_reduce', C.c_int32), ('reserved', C.c_int32)]


class DmcEstParams(C.Structure):
    _fields_ = [('num_modes', C.c_int32), ('ssf_pure', C.c_int32),
                ('ssf_pfw', C.c_int64), ('num_bins', C.c_int32),
                ('dens_pure', C.c_int32), ('dens_pfw', C.c_int64)]


# name -> (restype, argtypes); every symbol include/qmcwalk.h declares
SIGNATURES = {
    'qmc_last_error': (C.c_char_p, []),
    'qmc_abi_version': (C.c_int, []),
    'qmc_source_hash': (C.c_char_p, []),
    'qmc_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'qmc_model_one_body_table_info': (C.c_int, [C.POINTER(ModelParams),
                                                C.POINTER(C.c_int32),
                                                C.POINTER(C.c_int32), _dp]),
    'qmc_model_trig_table_info': (C.c_int, [C.POINTER(ModelParams),
                                            C.POINTER(C.c_int32), _dp]),
    'qmc_log_table_info': (C.c_int, [C.POINTER(C.c_int32), _dp]),
    'qmc_engine_create': (C.c_int, [C.POINTER(ModelParams), C.c_int, _vp,
                                    C.POINTER(_vp)]),
    'qmc_engine_create_on_stream': (C.c_int, [C.POINTER(ModelParams), C.c_int,
                                              _vp, C.POINTER(_vp)]),
    'qmc_engine_destroy': (None, [_vp]),
    'qmc_engine_set_fast_math': (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int)]),
    'qmc_engine_stream': (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_int)]),
    'qmc_engine_profile_begin': (C.c_int, [_vp, C.c_int64]),
    'qmc_engine_profile_end': (C.c_int, [_vp, _i64p, _dp, _dp, _dp]),
    'qmc_engine_section_profile': (C.c_int, [_vp, C.POINTER(C.c_uint64),
                                             C.POINTER(C.c_uint64), C.c_int32,
                                             C.c_int32]),
    'qmc_section_name': (C.c_char_p, [C.c_int32]),
    'qmc_engine_section_cut': (C.c_int, [_vp, C.c_int32]),
    'qmc_engine_diag_counters': (C.c_int, [_vp, C.POINTER(C.c_uint64),
                                           C.c_int32, C.c_int32]),
    'qmc_engine_sync': (C.c_int, [_vp]),
    'qmc_engine_timer_start': (C.c_int, [_vp]),
    'qmc_engine_timer_stop': (C.c_int, [_vp, C.POINTER(C.c_float)]),
    'qmc_evaluate': (C.c_int, [_vp, C.c_int64, _dp, _dp, _dp, _dp, _dp]),
    'qmc_evaluate_dev': (C.c_int, [_vp, C.c_int64, _vp, _vp, _vp, _vp, _vp]),
    'qmc_buffer_alloc': (C.c_int, [C.c_int, C.c_size_t, C.POINTER(_vp)]),
    'qmc_buffer_free': (C.c_int, [_vp]),
    'qmc_buffer_upload': (C.c_int, [_vp, _vp, C.c_size_t]),
    'qmc_buffer_download': (C.c_int, [_vp, _vp, C.c_size_t]),
    'qmc_vmc_create': (C.c_int, [_vp, C.POINTER(VmcParams), C.POINTER(_vp)]),
    'qmc_vmc_destroy': (None, [_vp]),
    'qmc_vmc_set_state': (C.c_int, [_vp, _dp]),
    'qmc_vmc_get_state': (C.c_int, [_vp, _dp, _dp, _dp]),
    'qmc_vmc_ssf': (C.c_int, [_vp, C.c_int32, _dp]),
    'qmc_vmc_run_block': (C.c_int, [_vp, C.c_int64, _dp, _dp, _i64p, _dp, _dp,
                                    _u8p, _dp]),
    'qmc_vmc_state_dev': (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    'qmc_vmc_block_sums_dev': (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp),
                                         C.POINTER(_vp)]),
    'qmc_vmc_set_tape': (C.c_int, [_vp, _dp, C.c_int64]),
    'qmc_dmc_create': (C.c_int, [_vp, C.POINTER(DmcParams), C.POINTER(_vp)]),
    'qmc_dmc_destroy': (None, [_vp]),
    'qmc_dmc_set_state': (C.c_int, [_vp, C.c_int64, _dp, C.c_int, C.c_double]),
    'qmc_dmc_set_state_dev': (C.c_int, [_vp, C.c_int64, _vp, C.c_int,
                                        C.c_double]),
    'qmc_dmc_set_state_from_vmc': (C.c_int, [_vp, _vp, C.c_int64, C.c_int,
                                             C.c_double]),
    'qmc_dmc_set_full_state': (C.c_int, [_vp, C.c_int64, _dp, _dp, _dp, _dp,
                                         C.c_double]),
    'qmc_dmc_run_block': (C.c_int, [_vp, C.c_int64, _dp, _dp, _u64p, _dp,
                                    _dp]),
    'qmc_dmc_set_estimators': (C.c_int, [_vp, C.POINTER(DmcEstParams)]),
    'qmc_dmc_run_block_est': (C.c_int, [_vp, C.c_int64, C.c_int, _dp, _dp,
                                        _u64p, _dp, _dp, _dp, _dp]),
    'qmc_dmc_est_begin_block': (C.c_int, [_vp, C.c_int64]),
    'qmc_dmc_step_estimators': (C.c_int, [_vp, C.c_int64]),
    'qmc_dmc_est_iter_dev': (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    'qmc_dmc_get_state': (C.c_int, [_vp, _dp, _dp, _dp, _u8p, _i64p, _dp]),
    'qmc_dmc_step_local': (C.c_int, [_vp, _vp]),
    'qmc_dmc_step_finish': (C.c_int, [_vp, _vp]),
    'qmc_dmc_read_series': (C.c_int, [_vp, C.c_int64, _dp, _dp, _u64p, _dp,
                                      _dp]),
    'qmc_dmc_num_walkers': (C.c_int, [_vp, _i64p]),
    'qmc_dmc_export_walkers': (C.c_int, [_vp, C.c_int64, C.c_int64, _vp]),
    'qmc_dmc_import_walkers': (C.c_int, [_vp, C.c_int64, _vp]),
    'qmc_dmc_walker_record_size': (C.c_int, [_vp, _i64p]),
    'qmc_dmc_import_walkers_at': (C.c_int, [_vp, C.c_int64, C.c_int64, _vp]),
    'qmc_dmc_set_num_walkers': (C.c_int, [_vp, C.c_int64]),
    'qmc_dmc_truncate': (C.c_int, [_vp, C.c_int64]),
    'qmc_dmc_set_tape': (C.c_int, [_vp, _dp, C.c_int64, _dp, C.c_int64, _i64p,
                                   _i64p, C.c_int64]),
}

_lib = None


def _preload_hip_runtime():
    """One HIP/HSA runtime per process.  PyTorch-ROCm wheels bundle their own
    libamdhip64.so (soname libamdhip64.so.7, the same as /opt/rocm's); two
    copies in one process fight over the KFD device ("no ROCm-capable device").
    Loading torch's copy first makes the dynamic loader satisfy our DT_NEEDED
    `libamdhip64.so.7` with it, so torch.distributed/RCCL and this engine
    share streams, events and device memory."""
    import importlib.util
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    cand = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
    if os.path.exists(cand):
        return C.CDLL(cand, mode=C.RTLD_GLOBAL)
    return None


def load():
    """Load libqmcwalk.so and attach the signatures; raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QmcError(
            f"{LIB_PATH} not found: the HIP engine has not been built "
            f"(run `make -C {os.path.join(_HERE, 'csrc')}`). "
            f"phd_qmclib_amd has no CPU fallback.")
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if a symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.qmc_abi_version() != 1:
        raise QmcError('libqmcwalk.so ABI version mismatch')
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().qmc_last_error()
        raise QmcError(msg.decode() if msg else f'libqmcwalk error {rc}')


def ptr(a, typ=_dp):
    """numpy array (or None) -> typed pointer."""
    return None if a is None else a.ctypes.data_as(typ)


def source_hash() -> str:
    """Identity of the kernels the loaded library was built from
    (qmc_source_hash: sha256 over csrc/ sources + flags, 16 hex digits)."""
    return load().qmc_source_hash().decode()


def device_count():
    n = C.c_int(0)
    check(load().qmc_device_count(C.byref(n)))
    return n.value
