"""Block-loop drivers shared by the VMC and DMC procedures.

The host loop of the reference's `Proc.exec` (qmc_exec/vmc/proc.py:87-250,
qmc_exec/dmc/proc.py:136-415): pull `SamplingBlock`s from
`sampling.blocks(...)`, discard the burn-in blocks, reduce every kept block to
its totals, wrap the totals in the reblocking containers.  It stays in Python:
one iteration per block, all the work is in the device block call.
"""
import typing as t
from itertools import islice

import numpy as np

from . import exec_logger
from .data import dmc as dmc_data, vmc as vmc_data
from ..qmc_base import dmc as dmc_base, vmc as vmc_base

__all__ = ['ProcInputError', 'exec_dmc', 'exec_vmc']


class ProcInputError(ValueError):
    """Flags an invalid input for a calculation procedure."""


def exec_vmc(proc, proc_input):
    """qmc_exec/vmc/proc.py:87-250."""
    from ..mrbp_qmc.vmc_exec import ProcInput
    num_blocks, ns = proc.num_blocks, proc.num_steps_block
    keep = proc.keep_iter_data
    exec_logger.info('Starting VMC sampling...')
    if not isinstance(proc_input, ProcInput):
        raise ProcInputError('the input data for the VMC procedure is '
                             'not valid')
    blocks_iter = proc.sampling.blocks(ns, proc_input.state)
    burn = proc.burn_in_blocks
    if burn is None:
        burn = num_blocks // 8
    block = None
    try:
        for block in islice(blocks_iter, burn):
            pass
        shape = (num_blocks, ns) if keep else (num_blocks,)
        wf = np.zeros(shape)
        en = np.zeros(shape)
        ssf = None
        if proc.should_eval_ssf:
            nm = proc.ssf_spec.num_modes
            ssf = np.zeros(((num_blocks, ns, nm, 3) if keep
                            else (num_blocks, nm, 3)))
        for b, block in enumerate(islice(blocks_iter, num_blocks)):
            p = block.iter_props
            if keep:
                wf[b], en[b] = p.wf_abs_log, p.energy
            else:
                wf[b], en[b] = p.wf_abs_log.mean(), p.energy.mean()
            if ssf is not None:
                ssf[b] = block.iter_ssf if keep else block.iter_ssf.mean(axis=0)
    finally:
        blocks_iter.close()
    exec_logger.info('VMC Sampling completed.')
    last_state = None if block is None else block.last_state
    props = vmc_base.PropsData(wf, en, np.zeros(shape, dtype=bool))
    energy_blocks = vmc_data.EnergyBlocks.from_data(props, bool(keep))
    ssf_blocks = None
    if ssf is not None:
        ssf_blocks = vmc_data.SSFBlocks.from_data(ssf, bool(keep))
    data = vmc_data.SamplingData(
        vmc_data.PropsDataBlocks(energy_blocks, ssf_blocks),
        (props, ssf) if keep else None)
    return proc.build_result(last_state, data)


def exec_dmc(proc, proc_input):
    """qmc_exec/dmc/proc.py:136-415: energy / weight / walkers series and the
    density / S(k) estimators of the kept blocks."""
    from ..mrbp_qmc.dmc_exec import ProcInput
    num_blocks, nts = proc.num_blocks, proc.num_time_steps_block
    keep = proc.keep_iter_data
    exec_logger.info('Starting DMC sampling...')
    burn = proc.burn_in_blocks
    if burn is None:
        burn = num_blocks // 8
    if not isinstance(proc_input, ProcInput):
        raise ProcInputError('the input data for the DMC procedure is '
                             'not valid')
    dens_spec, ssf_spec = proc.density_spec, proc.ssf_spec
    blocks_iter = proc.sampling.blocks(proc_input.state, nts, burn)
    block = None
    try:
        for block in islice(blocks_iter, burn):
            pass
        shape = (num_blocks, nts) if keep else (num_blocks,)
        e, w = np.zeros(shape), np.zeros(shape)
        nw = np.zeros(shape, dtype=np.uint64)
        re, ae = np.zeros(shape), np.zeros(shape)
        dens = ssf = None
        if dens_spec is not None:
            nb = dens_spec.num_bins
            dens = np.zeros((num_blocks, nts, nb, 1) if keep
                            else (num_blocks, nb, 1))
        if ssf_spec is not None:
            nm = ssf_spec.num_modes
            ssf = np.zeros((num_blocks, nts, nm, 3) if keep
                           else (num_blocks, nm, 3))
        pure_fac = np.ones(num_blocks)
        for b, block in enumerate(islice(blocks_iter, num_blocks)):
            p = block.iter_props
            if keep:
                e[b], w[b], nw[b] = p.energy, p.weight, p.num_walkers
                re[b], ae[b] = p.ref_energy, p.accum_energy
                if dens is not None:
                    dens[b] = block.iter_density
                if ssf is not None:
                    ssf[b] = block.iter_ssf
            else:
                wsum = p.weight.sum()
                e[b], w[b] = p.energy.sum(), wsum
                nw[b] = p.num_walkers.sum()
                re[b], ae[b] = p.ref_energy[-1], p.accum_energy[-1]
                pure_fac[b] = p.num_walkers[nts - 1] / wsum
                if dens is not None:
                    dens[b] = (block.iter_density[nts - 1]
                               if dens_spec.as_pure_est
                               else block.iter_density.sum(axis=0))
                if ssf is not None:
                    ssf[b] = (block.iter_ssf[nts - 1] if ssf_spec.as_pure_est
                              else block.iter_ssf.sum(axis=0))
    finally:
        blocks_iter.close()
    exec_logger.info('DMC sampling completed.')
    last_state = None if block is None else block.last_state
    props = dmc_base.PropsData(e, w, nw, re, ae)
    reduce_data = bool(keep)
    dens_blocks = ssf_blocks = None
    if dens is not None:
        dens_blocks = dmc_data.DensityBlocks.from_data(
            nts, dens[..., 0], props, reduce_data, dens_spec.as_pure_est,
            pure_fac)
    if ssf is not None:
        ssf_blocks = dmc_data.SSFBlocks.from_data(
            nts, ssf, props, reduce_data, ssf_spec.as_pure_est, pure_fac)
    blocks = dmc_data.PropsDataBlocks(
        dmc_data.EnergyBlocks.from_data(props, reduce_data),
        dmc_data.WeightBlocks.from_data(props, reduce_data),
        dmc_data.NumWalkersBlocks.from_data(props, reduce_data),
        dens_blocks, ssf_blocks)
    data = dmc_data.SamplingData(
        blocks, dmc_data.PropsDataSeries(props, ssf) if keep else None)
    return proc.build_result(last_state, data)
