"""Generic VMC / DMC sampling data contracts (reference: qmc_base/)."""
from . import dmc, vmc  # noqa: F401
