"""Statistics of serially correlated Monte-Carlo series (reference: stats/)."""
from . import reblock  # noqa: F401
