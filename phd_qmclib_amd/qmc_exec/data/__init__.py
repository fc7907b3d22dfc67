"""Block containers of a sampling run (reference: qmc_exec/data/)."""
from . import dmc, vmc  # noqa: F401
