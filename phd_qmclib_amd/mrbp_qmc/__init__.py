"""Bloch-Phonon multi-rods model (reference: mrbp_qmc/__init__.py)."""
from .model import (  # noqa: F401
    CFCSpec, CSWFOptimizer, OBFParams, Params, Spec, TBFParams, core_funcs,
    DIST_RAND, DIST_REGULAR, SysConfSlot
)
from . import dmc, vmc  # noqa: F401,E402
from . import dmc_exec, vmc_exec, wf_opt  # noqa: F401,E402
