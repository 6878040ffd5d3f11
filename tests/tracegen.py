"""Trace generators used by the tests (kept under this name for the test modules): they live in the package
(olavm_amd/air/tracegen.py) because bench.py builds its proof instance from the same padding rules."""
from olavm_amd.air.tracegen import *  # noqa: F401,F403
from olavm_amd.air.tracegen import P, T  # noqa: F401
