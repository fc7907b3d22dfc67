"""MI355X-native VMC/DMC walker-propagation engine for the PhD-QMCLib
`mrbp_qmc` sampling hot path (HIP/CDNA4 kernels behind a C-ABI; Python host
surface mirroring `qmc_base` / `mrbp_qmc` / `qmc_exec` / `stats`)."""
from . import constants, ideal  # noqa: F401

__version__ = '0.1.0'
