"""Identity stand-in for numba decorators (un-jitted execution of the reference)."""
def _ident(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    def deco(f):
        return f
    return deco
jit = njit = _ident
prange = range
def guvectorize(*a, **k):
    def deco(f):
        return f
    return deco
from . import config, runtime
