"""Model spec of the multi-rods Bloch-Phonon Bijl-Jastrow trial state.

Host-side mirror of the reference `mrbp_qmc.Spec` plugin surface
(reference: mrbp_qmc/model.py:134-400): the same constructor fields, defaults,
validation errors and derived-parameter tuples (`Params` :40-54, `OBFParams`
:57-65, `TBFParams` :68-75, `CFCSpec` :78-83), so an existing model definition
drops in unchanged.  The per-configuration functions (`wf_abs_log`, `energy`,
`drift`, `ith_energy_and_drift`) are served by the HIP engine through
`core_funcs` -- there is no CPU implementation in the product path.

Everything here runs once per `Spec`; it is double-precision scalar Python.
"""
import typing as t
from math import atan, ceil, cos, fabs, pi, sin, sqrt, tan

import attr
import numpy as np
from scipy.optimize import brentq

from .. import ideal

__all__ = ['CSWFOptimizer', 'CFCSpec', 'OBFParams', 'Params', 'Spec', 'TBFParams',
           'SysConfSlot', 'SysConfDistType', 'DIST_RAND', 'DIST_REGULAR',
           'core_funcs']

import enum


@enum.unique
class SysConfSlot(enum.IntEnum):
    """Rows of a system configuration `sys_conf[2, N]`
    (reference: qmc_base/jastrow/model.py:31-38)."""
    pos = 0
    drift = 1


class SysConfDistType(enum.Enum):
    """Initial particle arrangements (qmc_base/jastrow/model.py:41-44)."""
    RANDOM = 'random'
    REGULAR = 'regular'


DIST_RAND = SysConfDistType.RANDOM
DIST_REGULAR = SysConfDistType.REGULAR


class Params(t.NamedTuple):
    """The model `Spec` flattened (mrbp_qmc/model.py:40-54)."""
    lattice_depth: float
    lattice_ratio: float
    interaction_strength: float
    boson_number: int
    supercell_size: float
    tbf_contact_cutoff: float
    defect_magnitude: float
    defects_sep: int
    well_width: float
    barrier_width: float
    is_free: bool
    is_ideal: bool


class OBFParams(t.NamedTuple):
    """One-body (Kronig-Penney) factor parameters (mrbp_qmc/model.py:57-65)."""
    lattice_depth: float
    lattice_ratio: float
    well_width: float
    barrier_width: float
    param_e0: float
    param_k1: float
    param_kp1: float


class TBFParams(t.NamedTuple):
    """Two-body (phonon-matched) factor parameters (mrbp_qmc/model.py:68-75)."""
    supercell_size: float
    tbf_contact_cutoff: float
    param_k2: float
    param_beta: float
    param_r_off: float
    param_am: float


class CFCSpec(t.NamedTuple):
    """What the core functions need (mrbp_qmc/model.py:78-83)."""
    model_params: Params
    obf_params: OBFParams
    tbf_params: TBFParams


def _as_int(value):
    if isinstance(value, (int, np.integer)) and not isinstance(value, bool):
        return int(value)
    return value


def _opt(fn):
    return lambda v: None if v is None else fn(v)


def _check_cutoff(inst, attribute, value):
    # mrbp_qmc/model.py:95-107
    if not fabs(value) <= fabs(inst.supercell_size / 2):
        raise ValueError("parameter value 'rm' out of domain")


def _check_num_defects(inst, attribute, value):
    # mrbp_qmc/model.py:111-129
    if value is None:
        return
    if not isinstance(value, int):
        raise TypeError("'num_defects' must be an int")
    if value < 0:
        raise ValueError("number of defects can't be negative")
    num_sites = int(ceil(inst.supercell_size))
    if value and (num_sites % value):
        raise ValueError(f"the specified number of defects ({value:d}) "
                         f"can't be evenly distributed in the lattice")


@attr.s(auto_attribs=True, frozen=True)
class Spec:
    """Parameters of a 1D Bose gas in a multi-rods lattice with contact
    repulsion and a Bijl-Jastrow trial wave function
    (mrbp_qmc/model.py:134-171)."""

    #: Barrier height V0 of the Kronig-Penney lattice.
    lattice_depth: float = attr.ib(converter=float)
    #: Barrier width / well width.
    lattice_ratio: float = attr.ib(converter=float)
    #: Contact interaction strength g.
    interaction_strength: float = attr.ib(converter=float)
    #: Number of bosons N.
    boson_number: int = attr.ib(converter=_as_int,
                                validator=attr.validators.instance_of(int))
    #: Length L of the periodic simulation box (lattice periods).
    supercell_size: float = attr.ib(converter=float)
    #: Matching distance rm of the two-body factor (variational parameter).
    tbf_contact_cutoff: float = attr.ib(converter=float,
                                        validator=_check_cutoff)
    #: Number of evenly spaced lattice defects.
    num_defects: t.Optional[int] = attr.ib(default=None,
                                           converter=_opt(_as_int),
                                           validator=_check_num_defects)
    #: Barrier height inside a defect cell.
    defect_magnitude: t.Optional[float] = attr.ib(default=None,
                                                  converter=_opt(float))

    def __attrs_post_init__(self):
        # Defaults of the defect fields (mrbp_qmc/model.py:174-196).
        v0 = self.lattice_depth
        nd, dm = self.num_defects, self.defect_magnitude
        if nd is None and dm is None:
            nd, dm = 0, v0
        elif nd is None:
            nd, dm = 0, v0
        else:
            dm = dm if nd else v0
            if dm is None:
                # the reference compares None > float here and raises
                raise TypeError("'defect_magnitude' is required when "
                                "'num_defects' is nonzero")
            if dm > v0:
                raise ValueError("Defect magnitude can't be greater than "
                                 "the lattice depth.")
        object.__setattr__(self, 'num_defects', nd)
        object.__setattr__(self, 'defect_magnitude', dm)

    # -- geometry ---------------------------------------------------------
    @property
    def boundaries(self):
        return 0., 1. * self.supercell_size

    @property
    def well_width(self):
        return 1 / (1 + self.lattice_ratio)

    @property
    def barrier_width(self):
        return self.lattice_ratio / (1 + self.lattice_ratio)

    @property
    def is_free(self):
        """No external potential (mrbp_qmc/model.py:216-226)."""
        return self.lattice_depth <= 1e-10 or self.lattice_ratio <= 1e-10

    @property
    def is_ideal(self):
        """No interactions (mrbp_qmc/model.py:228-235)."""
        return self.interaction_strength <= 1e-10

    @property
    def sys_conf_shape(self):
        return len(SysConfSlot), self.boson_number

    def get_sys_conf_buffer(self):
        return np.zeros(self.sys_conf_shape, dtype=np.float64)

    def init_get_sys_conf(self, dist_type=DIST_RAND, offset=None):
        """A configuration with random or regular positions
        (mrbp_qmc/model.py:248-273); uses `numpy.random` like the reference."""
        n, L = self.boson_number, self.supercell_size
        z_min, _ = self.boundaries
        offset = offset or 0.
        if dist_type is DIST_RAND:
            spread = L * np.random.random_sample(n)
        elif dist_type is DIST_REGULAR:
            spread = np.linspace(0, L, n, endpoint=False)
        else:
            raise ValueError("unrecognized '{}' dist_type".format(dist_type))
        sys_conf = self.get_sys_conf_buffer()
        sys_conf[SysConfSlot.pos, :] = z_min + (offset + spread) % L
        return sys_conf

    # -- derived parameters -----------------------------------------------
    @property
    def params(self):
        num_sites = int(ceil(self.supercell_size))
        nd = self.num_defects
        defects_sep = 1 if not nd else int(num_sites // nd)
        return Params(self.lattice_depth, self.lattice_ratio,
                      self.interaction_strength, self.boson_number,
                      self.supercell_size, self.tbf_contact_cutoff,
                      self.defect_magnitude, defects_sep, self.well_width,
                      self.barrier_width, self.is_free, self.is_ideal)

    @property
    def obf_params(self):
        """e0 = lowest band edge, k1 = sqrt(e0), kp1 = sqrt(V0 - e0)
        (mrbp_qmc/model.py:298-315)."""
        v0, r = self.lattice_depth, self.lattice_ratio
        e0 = float(ideal.eigen_energy(v0, r))
        return OBFParams(v0, r, self.well_width, self.barrier_width,
                         param_e0=e0, param_k1=sqrt(e0),
                         param_kp1=sqrt(v0 - e0))

    @property
    def tbf_params(self):
        """Match the short-range cos(k2 (r - r_off)) solution of the two-body
        problem to the phononic tail sin(pi r / L)^beta at r = rm
        (mrbp_qmc/model.py:318-393)."""
        g, n, L = self.interaction_strength, self.boson_number, \
            self.supercell_size
        rm = self.tbf_contact_cutoff
        if not fabs(rm) <= fabs(L / 2):
            raise ValueError("parameter value 'rm' out of domain")
        if g == 0:
            return TBFParams(L, rm, param_k2=0., param_beta=0.,
                             param_r_off=1 / 2 * L, param_am=1.0)

        gamma = 0.5 * (L / n) ** 2 * g          # Lieb-Liniger gamma
        a1d = 2.0 / (gamma * n)                 # 1D scattering length / L
        rm /= L                                 # box units from here on

        def beta_rm(u):
            if u == 0:
                return tan(pi * rm) / pi
            return (u / pi * (rm - u * a1d * tan(u)) * tan(pi * rm) /
                    (u * a1d + rm * tan(u)))

        def local_energy_mismatch(u, *_):
            b = beta_rm(u)
            return ((u * sin(pi * rm)) ** 2 + (pi * b * cos(pi * rm)) ** 2 -
                    pi ** 2 * b * rm)

        u = brentq(local_energy_mismatch, 0, pi / 2, args=(a1d,))
        b = beta_rm(u)
        k2 = u / rm
        k2r_off = atan(1 / (k2 * a1d))
        beta = b / rm
        r_off = k2r_off / k2
        am = sin(pi * rm) ** beta / cos(u - k2r_off)
        return TBFParams(L, self.tbf_contact_cutoff, param_k2=k2 / L,
                         param_beta=beta, param_r_off=r_off * L, param_am=am)

    @property
    def cfc_spec(self):
        return CFCSpec(self.params, self.obf_params, self.tbf_params)


class _CoreFuncs:
    """Per-configuration functions of the model, with the call signatures of
    the reference's `mrbp_qmc.core_funcs` (qmc_base/jastrow/model.py:298-366,
    476-564, 756-773, 793-854).  Each call runs the HIP pair-sum kernel on
    the current device through the C-ABI; the engine for a given parameter
    set is created lazily and cached."""

    def __init__(self):
        self._engines = {}

    def _engine(self, model_params, obf_params, tbf_params):
        from ..engine import ModelEngine
        key = (tuple(model_params), tuple(obf_params), tuple(tbf_params))
        eng = self._engines.get(key)
        if eng is None:
            if len(self._engines) > 16:
                self._engines.clear()
            eng = ModelEngine(CFCSpec(model_params, obf_params, tbf_params))
            self._engines[key] = eng
        return eng

    def _eval(self, sys_conf, model_params, obf_params, tbf_params):
        sys_conf = np.asarray(sys_conf, dtype=np.float64)
        eng = self._engine(model_params, obf_params, tbf_params)
        return eng.evaluate(sys_conf[SysConfSlot.pos][np.newaxis, :])

    def wf_abs_log(self, sys_conf, model_params, obf_params, tbf_params):
        return float(self._eval(sys_conf, model_params, obf_params,
                                tbf_params).wf_abs_log[0])

    def energy(self, sys_conf, model_params, obf_params, tbf_params):
        return float(self._eval(sys_conf, model_params, obf_params,
                                tbf_params).energy[0])

    def drift(self, sys_conf, model_params, obf_params, tbf_params,
              result=None):
        out = self._eval(sys_conf, model_params, obf_params, tbf_params)
        if result is None:
            result = np.zeros_like(np.asarray(sys_conf, dtype=np.float64))
        result[SysConfSlot.pos, :] = np.asarray(sys_conf)[SysConfSlot.pos, :]
        result[SysConfSlot.drift, :] = out.drift[0]
        return result

    def ith_energy_and_drift(self, i_, sys_conf, model_params, obf_params,
                             tbf_params):
        out = self._eval(sys_conf, model_params, obf_params, tbf_params)
        return float(out.ith_energy[0, i_]), float(out.drift[0, i_])

    def ith_energy(self, i_, sys_conf, model_params, obf_params, tbf_params):
        return self.ith_energy_and_drift(i_, sys_conf, model_params,
                                         obf_params, tbf_params)[0]

    def ith_drift(self, i_, sys_conf, model_params, obf_params, tbf_params):
        return self.ith_energy_and_drift(i_, sys_conf, model_params,
                                         obf_params, tbf_params)[1]


core_funcs = _CoreFuncs()


class CSWFOptimizer:
    """Correlated-sampling optimisation of the trial wave function: finds the
    `tbf_contact_cutoff` that minimises the (re-weighted) variance of the
    local energy over a fixed set of configurations (reference:
    mrbp_qmc/model.py:818-942, qmc_base/jastrow/model.py:1125-1206).

    The configuration set is uploaded to HBM once; every trial value of the
    variational parameter is one batched launch of the pair-sum kernel
    (wf_abs_log + energy of all configurations) with that parameter set's
    engine.  The reference's `use_threads` / `num_workers` (dask scheduler
    knobs) are accepted and ignored."""

    def __init__(self, spec: Spec, sys_conf_set, ini_wf_abs_log_set,
                 ref_energy: t.Optional[float] = None,
                 use_threads: bool = True,
                 num_workers: t.Optional[int] = None,
                 verbose: bool = False):
        self.spec = spec
        self.sys_conf_set = np.asarray(sys_conf_set, dtype=np.float64)
        self.ini_wf_abs_log_set = np.asarray(ini_wf_abs_log_set,
                                             dtype=np.float64)
        self.ref_energy = ref_energy
        self.use_threads = use_threads
        self.num_workers = num_workers
        self.verbose = verbose
        scs = self.sys_conf_set
        n = spec.boson_number
        if scs.ndim == 3 and scs.shape[1:] == (len(SysConfSlot), n):
            pos = scs[:, SysConfSlot.pos, :]
        elif scs.ndim == 2 and scs.shape[1] == n:
            pos = scs
        else:
            raise ValueError('sys_conf_set must have shape (nconf, 2, '
                             'boson_number) or (nconf, boson_number)')
        if self.ini_wf_abs_log_set.shape != (pos.shape[0],):
            raise ValueError('ini_wf_abs_log_set must have one value per '
                             'configuration')
        # wf_abs_log and energy are symmetric in the particles: sorted
        # positions make the short-range branch of the kernel wave-uniform
        self._pos = np.sort(pos, axis=1)
        self._dev = None

    # -- device residency ---------------------------------------------
    def _buffers(self):
        if self._dev is None:
            from ..engine import DeviceBuffer
            W, n = self._pos.shape
            self._dev = (DeviceBuffer((W, n)).upload(self._pos),
                         DeviceBuffer((W,)), DeviceBuffer((W,)))
        return self._dev

    def close(self):
        if self._dev is not None:
            for b in self._dev:
                b.close()
            self._dev = None

    def update_spec(self, tbf_contact_cutoff: float):
        """The model spec with a new value of the variational parameter."""
        return attr.evolve(self.spec,
                           tbf_contact_cutoff=float(tbf_contact_cutoff))

    def wf_abs_log_and_energy_set(self, cfc_spec: CFCSpec):
        """wf_abs_log and local energy of every configuration of the set
        under the given parameters."""
        from ..engine import ModelEngine
        pos_d, wf_d, en_d = self._buffers()
        eng = ModelEngine(cfc_spec, device=pos_d.device)
        try:
            eng.evaluate_dev(self._pos.shape[0], pos_d.ptr, wf_d.ptr,
                             en_d.ptr)
            eng.sync()
        finally:
            eng.close()
        return wf_d.download(), en_d.download()

    @staticmethod
    def weighed_variance(weights_log_set, energy_set, ref_energy=None):
        """Weighed variance of the energies
        (qmc_base/jastrow/model.py:1146-1164; the reference ignores
        `ref_energy` and uses the weighed mean)."""
        rel_weights = np.exp(weights_log_set - weights_log_set.max())
        weight_sum = rel_weights.sum()
        ref_energy = (rel_weights * energy_set).sum() / weight_sum
        e_diff = rel_weights * (energy_set - ref_energy) ** 2
        return e_diff.sum() / weight_sum

    def principal_function(self, tbf_contact_cutoff):
        """The weighed variance of the local energy for a trial cutoff."""
        cutoff = float(np.ravel(tbf_contact_cutoff)[0])
        cfc_spec = self.update_spec(cutoff).cfc_spec
        wf_set, energy_set = self.wf_abs_log_and_energy_set(cfc_spec)
        weights_log_set = 2 * (wf_set - self.ini_wf_abs_log_set)
        return self.weighed_variance(weights_log_set, energy_set)

    @property
    def principal_function_bounds(self):
        sc_size = self.spec.supercell_size
        return [(5e-2, (0.5 - 5e-3) * sc_size)]

    def exec(self, seed=None):
        """Runs the minimisation (scipy `differential_evolution`, as the
        reference) and returns the spec with the optimal cutoff."""
        from scipy.optimize import differential_evolution
        try:
            res = differential_evolution(self.principal_function,
                                         bounds=self.principal_function_bounds,
                                         disp=self.verbose, seed=seed)
        finally:
            self.close()
        opt_cutoff, = res.x
        return self.update_spec(opt_cutoff)
