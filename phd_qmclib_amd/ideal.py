"""Ground-state band edge of the Kronig-Penney (multi-rods) lattice.

Host-side, evaluated once per `Spec`.  Mirrors the behaviour of the reference
`ideal.eigen_energy` (ideal.py:57-85): a machine-precision bracketing root
(`scipy.optimize.brentq`) polished by `mpmath.findroot`; the pure-mpmath
Illinois path is taken when the double-precision relation overflows.
"""
import math
from functools import partial

import mpmath as mp
from scipy.optimize import brentq

__all__ = ['band_edge_relation', 'eigen_energy']


def band_edge_relation(v0, r, energy, momentum=0.0, ctx=math):
    """Kronig-Penney dispersion relation, F(E) - cos(k_s) (ideal.py:8-53).

    Lattice period 1, barrier height ``v0``, barrier/well width ratio ``r``:
    well width a = 1/(1+r), barrier width b = r/(1+r).  With k = sqrt(E) and
    kappa = sqrt(v0 - E),

        F(E) = cosh(kappa b) cos(k a)
               + (kappa^2 - k^2) / (2 kappa k) sinh(kappa b) sin(k a).

    The E -> 0 and E -> v0 limits are taken analytically.
    """
    a = 1 / (1 + r)
    b = r / (1 + r)
    if energy == 0:
        rv = ctx.sqrt(v0)
        return (1 / (2 * (1 + r)) * rv * ctx.sinh(b * rv)
                + ctx.cosh(b * rv) - ctx.cos(momentum))
    if energy == v0:
        rv = ctx.sqrt(v0)
        return (-r * rv / (2 * (1 + r)) * ctx.sin(rv * a)
                + ctx.cos(rv * a) - ctx.cos(momentum))
    kin = ctx.sqrt(energy)
    kap = ctx.sqrt(v0 - energy)
    return ((v0 - 2 * energy) / (2 * ctx.sqrt(energy * (v0 - energy)))
            * ctx.sinh(b * kap) * ctx.sin(kin * a)
            + ctx.cosh(b * kap) * ctx.cos(kin * a) - ctx.cos(momentum))


def eigen_energy(lattice_depth, lattice_ratio):
    """Lowest band-edge energy e0 of the lattice (ideal.py:57-85)."""
    v0, r = lattice_depth, lattice_ratio
    upper = min(v0, (1 + r) ** 2 * math.pi ** 2)
    try:
        root = brentq(partial(band_edge_relation, v0, r, momentum=0), 0, upper)
        solver = partial(mp.findroot, verify=False)
    except OverflowError:
        root = (0, min(v0, (1 + r) ** 2 * mp.pi ** 2))
        solver = partial(mp.findroot, solver='illinois', verify=False)
    root = solver(partial(band_edge_relation, v0, r, momentum=0, ctx=mp), root)
    return mp.chop(root)
