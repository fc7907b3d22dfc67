"""Import harness for running the reference (PhD-QMCLib) un-jitted.

TEST INFRASTRUCTURE ONLY.  This file is used in the build container, where
`/root/reference` exists, to *generate* the golden vectors committed under
`tests/golden/`.  Nothing here is imported by the product package, by the
`-m gpu` tests, by `bench.py` or by `__graft_entry__.smoke()`; it never runs
on the GPU box (the reference does not travel).

What it does:
  * numba is not importable in this image.  Every `@jit(nopython=True)` body of
    the reference is valid plain Python, so `shim/numba` provides identity
    decorators (`jit`, `njit`, `prange = range`) and the reference's function
    bodies run in CPython unchanged.  No reference source is copied.
  * the reference writes `class X(Base, typing.NamedTuple)` (python 3.7); on
    python >= 3.9 that needs the NamedTupleMeta patch below.
  * `np.int` / `np.bool` aliases removed from numpy >= 1.24 are restored.
  * trivial stubs for CLI-only dependencies (colorlog, tzlocal, dotenv,
    colored, ruamel.yaml) which `phd_qmclib.mrbp_qmc.__init__` imports.

Interpreter: /opt/conda/bin/python3.9 (has cached_property, h5py, dask, attrs,
mpmath, scipy).  Usage: `import harness` before importing `phd_qmclib`.
"""
import collections
import collections.abc
import os
import sys
import typing

import numpy as np

np.int = int
np.bool = np.bool_

for _n in ('Mapping', 'Sequence', 'MutableMapping', 'Iterable'):
    if not hasattr(collections, _n):
        setattr(collections, _n, getattr(collections.abc, _n))

_orig_new = typing.NamedTupleMeta.__new__


def _nt_new(cls, typename, bases, ns):
    return _orig_new(cls, typename, (typing._NamedTuple,), ns)


typing.NamedTupleMeta.__new__ = _nt_new
typing.NamedTuple.__mro_entries__ = lambda bases: (typing._NamedTuple,)

_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_here, 'shim'))
sys.path.insert(0, os.environ.get('QMC_REFERENCE_SRC', '/root/reference/src'))


class RNGTape:
    """Records every `numpy.random.rand()` / `numpy.random.normal()` draw the
    un-jitted reference makes (the names its bodies resolve at call time)."""

    def __init__(self):
        self.uniform = []
        self.normal = []
        self.kinds = []        # 'u' / 'n' per draw, in call order
        self._rand = np.random.rand
        self._normal = np.random.normal

    def __enter__(self):
        def rand(*a):
            assert not a
            v = self._rand()
            self.uniform.append(v)
            self.kinds.append('u')
            return v

        def normal(loc=0.0, scale=1.0, size=None):
            # numpy's legacy normal is `loc + scale * gauss()`; the tape keeps
            # the standard deviate so a replay does the same arithmetic.
            assert size is None
            g = self._normal()
            self.normal.append(g)
            self.kinds.append('n')
            return loc + scale * g

        np.random.rand = rand
        np.random.normal = normal
        return self

    def __exit__(self, *exc):
        np.random.rand = self._rand
        np.random.normal = self._normal
        return False
