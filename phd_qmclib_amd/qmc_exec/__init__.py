"""Batch drivers of the samplings (reference: qmc_exec/)."""
import logging

exec_logger = logging.getLogger('phd_qmclib_amd.exec')

from . import data  # noqa: E402,F401
