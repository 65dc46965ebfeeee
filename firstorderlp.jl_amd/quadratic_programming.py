"""Host-side data model mirroring src/quadratic_programming.jl of the reference.

``QuadraticProgrammingProblem`` (quadratic_programming.jl:34-76) keeps the same
field names; ``SparseMatrixCSC{Float64,Int64}`` becomes
``scipy.sparse.csc_matrix`` with float64 data, int64 indices and sorted row
indices -- the same three arrays (colptr, rowval, nzval), 0-based, that the
C-ABI ``pdhg_create`` ingests.
"""
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp


def as_csc(matrix, shape=None):
    """sparse(matrix) -> canonical CSC (float64, int64, sorted, no duplicates)."""
    if sp.issparse(matrix):
        a = sp.csc_matrix(matrix, dtype=np.float64, copy=True)
    else:
        a = sp.csc_matrix(np.asarray(matrix, dtype=np.float64))
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"matrix shape {a.shape} != {shape}")
    a.sum_duplicates()
    a.sort_indices()
    a.indices = a.indices.astype(np.int64, copy=False)
    a.indptr = a.indptr.astype(np.int64, copy=False)
    return a


@dataclass
class QuadraticProgrammingProblem:
    """minimize 1/2 x'Qx + c'x + objective_constant
    s.t. A[:num_equalities] x == b[:num_equalities],
         A[num_equalities:] x >= b[num_equalities:],  lb <= x <= ub.
    (quadratic_programming.jl:15-76)"""
    variable_lower_bound: np.ndarray
    variable_upper_bound: np.ndarray
    objective_matrix: sp.csc_matrix
    objective_vector: np.ndarray
    objective_constant: float
    constraint_matrix: sp.csc_matrix
    right_hand_side: np.ndarray
    num_equalities: int

    def __post_init__(self):
        f = lambda v: np.array(v, dtype=np.float64).reshape(-1)
        self.variable_lower_bound = f(self.variable_lower_bound)
        self.variable_upper_bound = f(self.variable_upper_bound)
        self.objective_vector = f(self.objective_vector)
        self.right_hand_side = f(self.right_hand_side)
        n = len(self.variable_lower_bound)
        m = len(self.right_hand_side)
        self.objective_matrix = as_csc(self.objective_matrix, (n, n))
        self.constraint_matrix = as_csc(self.constraint_matrix, (m, n))
        self.objective_constant = float(self.objective_constant)
        self.num_equalities = int(self.num_equalities)

    @property
    def num_variables(self):
        return len(self.variable_lower_bound)

    @property
    def num_constraints(self):
        return len(self.right_hand_side)

    def copy(self):
        return QuadraticProgrammingProblem(
            self.variable_lower_bound.copy(), self.variable_upper_bound.copy(),
            self.objective_matrix.copy(), self.objective_vector.copy(),
            self.objective_constant, self.constraint_matrix.copy(),
            self.right_hand_side.copy(), self.num_equalities)


def linear_programming_problem(variable_lower_bound, variable_upper_bound,
                               objective_vector, objective_constant,
                               constraint_matrix, right_hand_side,
                               num_equalities):
    """quadratic_programming.jl:255-277 (objective_matrix = spzeros(n, n))."""
    n = len(variable_lower_bound)
    return QuadraticProgrammingProblem(
        variable_lower_bound, variable_upper_bound,
        sp.csc_matrix((n, n), dtype=np.float64), objective_vector,
        objective_constant, constraint_matrix, right_hand_side, num_equalities)


def is_linear_programming_problem(problem):
    """quadratic_programming.jl:282-284: nnz(objective_matrix) == 0."""
    return problem.objective_matrix.nnz == 0


@dataclass
class ScaledQpProblem:
    """quadratic_programming.jl:293-298."""
    original_qp: QuadraticProgrammingProblem
    scaled_qp: QuadraticProgrammingProblem
    constraint_rescaling: np.ndarray
    variable_rescaling: np.ndarray


def equality_range(problem):
    """quadratic_programming.jl:300 (0-based slice)."""
    return slice(0, problem.num_equalities)


def inequality_range(problem):
    """quadratic_programming.jl:302-304 (0-based slice)."""
    return slice(problem.num_equalities, problem.num_constraints)
