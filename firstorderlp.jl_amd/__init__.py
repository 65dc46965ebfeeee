"""MI355X-native PDHG inner loop behind FirstOrderLp.jl's API surface.

Host side (this package) mirrors the reference's names --
``QuadraticProgrammingProblem``, ``PdhgParameters``, ``optimize`` ... -- and
drives hand-written HIP kernels for gfx950 through the C ABI declared in
``include/pdhg_hip.h`` (``csrc/libpdhg_hip.so``).
"""
from .quadratic_programming import (  # noqa: F401
    QuadraticProgrammingProblem, ScaledQpProblem, linear_programming_problem,
    is_linear_programming_problem, equality_range, inequality_range)
from .engine import HipPdhgEngine  # noqa: F401
from . import _lib  # noqa: F401
