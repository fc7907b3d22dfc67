"""Wave-function optimisation procedure (reference:
mrbp_qmc/wf_opt/wf_opt.py:14-66): the step a `Proc` runs between the VMC
sampling and the DMC run when asked to optimise the trial function."""
import typing as t

import attr
import numpy as np

from . import model

__all__ = ['WFOptProc']


def _opt_float(v):
    return None if v is None else float(v)


@attr.s(auto_attribs=True, frozen=True)
class WFOptProc:
    """Wave function optimization."""

    #: The number of configurations used in the process.
    num_sys_confs: int = attr.ib(default=1024,
                                 validator=attr.validators.instance_of(int))

    #: The energy of reference to minimize the variance of the local energy.
    ref_energy: t.Optional[float] = attr.ib(default=None, converter=_opt_float)

    #: Accepted for compatibility (the reference's dask scheduler knobs).
    use_threads: bool = attr.ib(default=True,
                                validator=attr.validators.instance_of(bool))
    num_workers: t.Optional[int] = attr.ib(
        default=None,
        validator=attr.validators.optional(attr.validators.instance_of(int)))

    #: Display log messages or not.
    verbose: bool = attr.ib(default=False,
                            validator=attr.validators.instance_of(bool))

    def exec(self, model_spec: model.Spec, sys_conf_set: np.ndarray,
             ini_wf_abs_log_set: np.ndarray, seed=None):
        """Minimises the variance over the last `num_sys_confs`
        configurations; returns the spec with the optimal cutoff."""
        num_sys_confs = self.num_sys_confs
        sys_conf_set = sys_conf_set[-num_sys_confs:]
        ini_wf_abs_log_set = ini_wf_abs_log_set[-num_sys_confs:]
        optimizer = model.CSWFOptimizer(model_spec, sys_conf_set,
                                        ini_wf_abs_log_set, self.ref_energy,
                                        self.use_threads, self.num_workers,
                                        self.verbose)
        return optimizer.exec(seed=seed)
