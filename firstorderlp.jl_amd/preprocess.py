"""One-shot host setup before the loop: validation and diagonal rescaling,
mirroring src/preprocess.jl (validate :18-84, l2_norm :99-113, l2_norm_rescaling
:358-372, ruiz_rescaling :412-477, pock_chambolle_rescaling :508-539,
scale_problem :555-573, rescale_problem :631-687)."""
import numpy as np
import scipy.sparse as sp

from .quadratic_programming import (QuadraticProgrammingProblem,
                                    ScaledQpProblem, as_csc)


def validate(p):
    """preprocess.jl:18-84: raises on a malformed problem."""
    errors = []
    n = len(p.objective_vector)
    if len(p.variable_lower_bound) != len(p.variable_upper_bound):
        errors.append("length(variable_lower_bound) != length(variable_upper_bound)")
    if len(p.variable_lower_bound) != n:
        errors.append("length(variable_lower_bound) != length(objective_vector)")
    if p.constraint_matrix.shape[0] != len(p.right_hand_side):
        errors.append("size(constraint_matrix, 1) != length(right_hand_side)")
    if p.constraint_matrix.shape[1] != n:
        errors.append("size(constraint_matrix, 2) != length(objective_vector)")
    if p.objective_matrix.shape != (n, n):
        errors.append("objective_matrix is not square with length(objective_vector)")
    if np.any(p.variable_lower_bound == np.inf):
        errors.append("variable_lower_bound contains +Inf")
    if np.any(p.variable_upper_bound == -np.inf):
        errors.append("variable_upper_bound contains -Inf")
    if np.any(np.isnan(p.variable_lower_bound)) or np.any(np.isnan(p.variable_upper_bound)):
        errors.append("NaN found in variable bounds")
    if not np.all(np.isfinite(p.right_hand_side)):
        errors.append("NaN or Inf found in right hand side")
    if not np.all(np.isfinite(p.objective_vector)):
        errors.append("NaN or Inf found in objective vector")
    if not np.all(np.isfinite(p.constraint_matrix.data)):
        errors.append("NaN or Inf found in constraint matrix")
    if not np.all(np.isfinite(p.objective_matrix.data)):
        errors.append("NaN or Inf found in objective matrix")
    if errors:
        raise ValueError("Error found when validating QuadraticProgrammingProblem: "
                         + "; ".join(errors))
    return True


class _CscView:
    """Row/column reductions and diagonal scaling directly on a CSC matrix's
    arrays (no sparse matmul, no reallocation): the stored pattern never
    changes under diagonal rescaling."""

    def __init__(self, matrix):
        self.A = matrix
        self.m, self.n = matrix.shape
        self.indptr = matrix.indptr
        self.indices = matrix.indices
        self.col_counts = np.diff(self.indptr)
        self._col_of_nz = None
        self._perm = None
        self._row_indptr = None

    @property
    def col_of_nz(self):
        if self._col_of_nz is None:
            self._col_of_nz = np.repeat(np.arange(self.n, dtype=np.int64), self.col_counts)
        return self._col_of_nz

    def _row_major(self):
        if self._perm is None:
            self._perm = np.argsort(self.indices, kind="stable")
            counts = np.bincount(self.indices, minlength=self.m)
            self._row_indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        return self._perm, self._row_indptr

    @staticmethod
    def _segments(values, indptr, ufunc, size):
        out = np.zeros(size)
        counts = np.diff(indptr)
        nonempty = counts > 0
        if len(values) and nonempty.any():
            red = ufunc.reduceat(values, indptr[:-1][nonempty])
            out[nonempty] = red
        return out

    def reduce(self, dims, values, ufunc):
        """Julia dims: 1 -> one result per column, 2 -> one per row."""
        if dims == 1:
            return self._segments(values, self.indptr, ufunc, self.n)
        perm, row_indptr = self._row_major()
        return self._segments(values[perm], row_indptr, ufunc, self.m)

    def scale(self, constraint_rescaling, variable_rescaling):
        """(Diagonal(1 ./ E) * A) * Diagonal(1 ./ D), entry by entry, in that order."""
        data = self.A.data
        if len(constraint_rescaling):
            data *= (1.0 / constraint_rescaling)[self.indices]
        data *= (1.0 / variable_rescaling)[self.col_of_nz]


def _view(matrix):
    v = getattr(matrix, "_folp_view", None)
    if v is None or v.A is not matrix:
        v = _CscView(matrix)
        matrix._folp_view = v
    return v


def _max_abs(matrix, dims):
    """vec(maximum(abs, matrix, dims=dims)) with Julia's dims (1: per column,
    2: per row); rows/columns without stored entries give 0."""
    return _view(matrix).reduce(dims, np.abs(matrix.data), np.maximum)


def _sum_f(matrix, dims, f):
    """vec(sum(f, matrix, dims=dims)) over stored entries (f(0) = 0 here)."""
    return _view(matrix).reduce(dims, f(matrix.data), np.add)


def l2_norm(matrix, dimension):
    """preprocess.jl:99-113: scaled sum of squares to avoid overflow."""
    v = _view(matrix)
    scale_factor = _max_abs(matrix, dimension)
    scale_factor[scale_factor == 0.0] = 1.0
    inv = 1.0 / scale_factor
    scaled = matrix.data * (inv[v.col_of_nz] if dimension == 1 else inv[v.indices])
    return scale_factor * np.sqrt(v.reduce(dimension, scaled * scaled, np.add))


def scale_problem(problem, constraint_rescaling, variable_rescaling):
    """preprocess.jl:555-573 (in place)."""
    assert np.all(constraint_rescaling > 0) and np.all(variable_rescaling > 0)
    problem.objective_vector = problem.objective_vector / variable_rescaling
    if problem.objective_matrix.nnz:
        dinv = sp.diags(1.0 / variable_rescaling)
        problem.objective_matrix = as_csc((dinv @ problem.objective_matrix) @ dinv)
    problem.variable_upper_bound = problem.variable_upper_bound * variable_rescaling
    problem.variable_lower_bound = problem.variable_lower_bound * variable_rescaling
    problem.right_hand_side = problem.right_hand_side / constraint_rescaling
    _view(problem.constraint_matrix).scale(constraint_rescaling, variable_rescaling)


def unscale_problem(problem, constraint_rescaling, variable_rescaling):
    """preprocess.jl:580-587"""
    scale_problem(problem, 1.0 / constraint_rescaling, 1.0 / variable_rescaling)


def l2_norm_rescaling(problem):
    """preprocess.jl:358-372"""
    norm_of_rows = l2_norm(problem.constraint_matrix, 2)
    norm_of_columns = l2_norm(problem.constraint_matrix, 1)
    norm_of_rows[norm_of_rows == 0.0] = 1.0
    norm_of_columns[norm_of_columns == 0.0] = 1.0
    column_rescale_factor = np.sqrt(norm_of_columns)
    row_rescale_factor = np.sqrt(norm_of_rows)
    scale_problem(problem, row_rescale_factor, column_rescale_factor)
    return row_rescale_factor, column_rescale_factor


def ruiz_rescaling(problem, num_iterations, p=np.inf):
    """preprocess.jl:412-477"""
    num_constraints, num_variables = problem.constraint_matrix.shape
    cum_constraint_rescaling = np.ones(num_constraints)
    cum_variable_rescaling = np.ones(num_variables)
    for _ in range(num_iterations):
        constraint_matrix = problem.constraint_matrix
        objective_matrix = problem.objective_matrix
        if p == np.inf:
            variable_rescaling = np.sqrt(np.maximum(_max_abs(constraint_matrix, 1),
                                                    _max_abs(objective_matrix, 1)))
        else:
            assert p == 2
            variable_rescaling = np.sqrt(np.sqrt(l2_norm(constraint_matrix, 1) ** 2 +
                                                 l2_norm(objective_matrix, 1) ** 2))
        variable_rescaling[variable_rescaling == 0.0] = 1.0
        if num_constraints == 0:
            constraint_rescaling = np.zeros(0)
        else:
            if p == np.inf:
                constraint_rescaling = np.sqrt(_max_abs(constraint_matrix, 2))
            else:
                norm_of_rows = l2_norm(problem.constraint_matrix, 2)
                if problem.objective_matrix.nnz == 0 or not np.any(problem.objective_matrix.data):
                    target_row_norm = np.sqrt(num_variables / num_constraints)
                else:
                    target_row_norm = np.sqrt(num_variables / (num_constraints + num_variables))
                constraint_rescaling = np.sqrt(norm_of_rows / target_row_norm)
            constraint_rescaling[constraint_rescaling == 0.0] = 1.0
        scale_problem(problem, constraint_rescaling, variable_rescaling)
        cum_constraint_rescaling *= constraint_rescaling
        cum_variable_rescaling *= variable_rescaling
    return cum_constraint_rescaling, cum_variable_rescaling


def pock_chambolle_rescaling(problem, alpha):
    """preprocess.jl:508-539"""
    assert 0 <= alpha <= 2
    constraint_matrix = problem.constraint_matrix
    m, n = constraint_matrix.shape
    v = _view(constraint_matrix)
    col_sum = _sum_f(constraint_matrix, 1, lambda t: np.abs(t) ** (2 - alpha))
    row_sum = _sum_f(constraint_matrix, 2, lambda t: np.abs(t) ** alpha)
    # Julia's mapreduce over a sparse matrix visits the structural zeros too,
    # and 0.0^0 == 1.0: with exponent 0 every absent entry contributes 1.
    if 2 - alpha == 0:
        col_sum = col_sum + (m - v.col_counts)
    if alpha == 0:
        row_sum = row_sum + (n - np.bincount(v.indices, minlength=m))
    variable_rescaling = np.sqrt(col_sum)
    constraint_rescaling = np.sqrt(row_sum)
    variable_rescaling[variable_rescaling == 0.0] = 1.0
    constraint_rescaling[constraint_rescaling == 0.0] = 1.0
    scale_problem(problem, constraint_rescaling, variable_rescaling)
    return constraint_rescaling, variable_rescaling


def rescale_problem(l_inf_ruiz_iterations, l2_norm_rescaling_flag,
                    pock_chambolle_alpha, verbosity, original_problem):
    """preprocess.jl:631-687: returns a ScaledQpProblem; original untouched."""
    problem = original_problem.copy()
    num_constraints, num_variables = problem.constraint_matrix.shape
    constraint_rescaling = np.ones(num_constraints)
    variable_rescaling = np.ones(num_variables)
    if l_inf_ruiz_iterations > 0:
        con_rescale, var_rescale = ruiz_rescaling(problem, l_inf_ruiz_iterations, np.inf)
        constraint_rescaling *= con_rescale
        variable_rescaling *= var_rescale
    if l2_norm_rescaling_flag:
        con_rescale, var_rescale = l2_norm_rescaling(problem)
        constraint_rescaling *= con_rescale
        variable_rescaling *= var_rescale
    if pock_chambolle_alpha is not None:
        con_rescale, var_rescale = pock_chambolle_rescaling(problem, pock_chambolle_alpha)
        constraint_rescaling *= con_rescale
        variable_rescaling *= var_rescale
    scaled_problem = ScaledQpProblem(original_problem, problem, constraint_rescaling,
                                     variable_rescaling)
    if verbosity >= 3:
        if l_inf_ruiz_iterations == 0 and not l2_norm_rescaling_flag:
            print("No rescaling.")
        else:
            print(f"Problem after rescaling (Ruiz iterations = {l_inf_ruiz_iterations}, "
                  f"l2_norm_rescaling = {l2_norm_rescaling_flag}):")
    return scaled_problem


# ---- presolve (preprocess.jl:122-340) ------------------------------------------

from dataclasses import dataclass as _dataclass


def remove_empty_rows(problem):
    """preprocess.jl:122-145 (in place); returns the 0-based empty row indices."""
    A = problem.constraint_matrix
    seen_row = np.zeros(A.shape[0], dtype=bool)
    seen_row[A.indices] = True
    empty_rows = np.nonzero(~seen_row)[0]
    for row in empty_rows:
        if row >= problem.num_equalities and problem.right_hand_side[row] > 0.0:
            raise ValueError("The problem is infeasible.")
        elif row < problem.num_equalities and problem.right_hand_side[row] != 0.0:
            raise ValueError("The problem is infeasible.")
    if len(empty_rows):
        problem.constraint_matrix = as_csc(sp.csr_matrix(A)[seen_row, :])
        problem.right_hand_side = problem.right_hand_side[seen_row]
        problem.num_equalities -= int(np.sum(empty_rows < problem.num_equalities))
    return empty_rows


def remove_empty_columns(problem):
    """preprocess.jl:156-186 (in place); LP only."""
    assert problem.objective_matrix.nnz == 0 or not np.any(problem.objective_matrix.data)
    A = problem.constraint_matrix
    is_empty_column = np.diff(A.indptr) == 0
    empty_columns = np.nonzero(is_empty_column)[0]
    if len(empty_columns) == 0:
        return empty_columns
    for col in empty_columns:
        coef = problem.objective_vector[col]
        bound = problem.variable_lower_bound[col] if coef >= 0 else problem.variable_upper_bound[col]
        problem.objective_constant += bound * coef
    keep = ~is_empty_column
    problem.constraint_matrix = as_csc(A[:, keep])
    problem.objective_vector = problem.objective_vector[keep]
    problem.variable_lower_bound = problem.variable_lower_bound[keep]
    problem.variable_upper_bound = problem.variable_upper_bound[keep]
    problem.objective_matrix = as_csc(problem.objective_matrix[keep, :][:, keep])
    return empty_columns


def transform_bounds_into_linear_constraints(qp):
    """preprocess.jl:191-222"""
    lo_idx = np.nonzero(np.isfinite(qp.variable_lower_bound))[0]
    up_idx = np.nonzero(np.isfinite(qp.variable_upper_bound))[0]
    k = len(lo_idx) + len(up_idx)
    block = sp.csc_matrix((np.concatenate([np.ones(len(lo_idx)), -np.ones(len(up_idx))]),
                           (np.arange(k), np.concatenate([lo_idx, up_idx]))),
                          shape=(k, len(qp.variable_lower_bound)))
    qp.constraint_matrix = as_csc(sp.vstack([qp.constraint_matrix, block]))
    qp.right_hand_side = np.concatenate([qp.right_hand_side, qp.variable_lower_bound[lo_idx],
                                         -qp.variable_upper_bound[up_idx]])
    qp.variable_lower_bound = np.full_like(qp.variable_lower_bound, -np.inf)
    qp.variable_upper_bound = np.full_like(qp.variable_upper_bound, np.inf)


@_dataclass
class PresolveInfo:
    """preprocess.jl:224-231"""
    original_primal_size: int
    original_dual_size: int
    empty_rows: np.ndarray
    empty_columns: np.ndarray
    variable_lower_bound: np.ndarray
    variable_upper_bound: np.ndarray


def presolve(qp, verbosity=1, transform_bounds=False):
    """preprocess.jl:236-268 (modifies qp in place)."""
    saved_lb = qp.variable_lower_bound.copy()
    saved_ub = qp.variable_upper_bound.copy()
    original_dual_size, original_primal_size = qp.constraint_matrix.shape
    empty_rows = remove_empty_rows(qp)
    if qp.objective_matrix.nnz == 0 or not np.any(qp.objective_matrix.data):
        empty_columns = remove_empty_columns(qp)
    else:
        empty_columns = np.zeros(0, dtype=np.int64)
    if verbosity >= 1:
        counts = np.bincount(qp.constraint_matrix.indices, minlength=qp.constraint_matrix.shape[0])
        num_single = int(np.sum(counts == 1))
        if num_single > 0:
            print(f"{num_single} constraints involving exactly a single variable")
    if transform_bounds:
        transform_bounds_into_linear_constraints(qp)
    return PresolveInfo(original_primal_size, original_dual_size, empty_rows, empty_columns,
                        saved_lb, saved_ub)


def recover_original_solution(solution, empty_indices, original_size):
    """preprocess.jl:294-307"""
    nonempty = np.ones(original_size, dtype=bool)
    nonempty[empty_indices] = False
    original = np.zeros(original_size)
    original[nonempty] = solution[:int(nonempty.sum())]
    return original


def undo_presolve(presolve_info, primal_solution, dual_solution):
    """preprocess.jl:309-332"""
    primal = recover_original_solution(primal_solution, presolve_info.empty_columns,
                                       presolve_info.original_primal_size)
    primal = np.minimum(presolve_info.variable_upper_bound,
                        np.maximum(presolve_info.variable_lower_bound, primal))
    dual = recover_original_solution(dual_solution, presolve_info.empty_rows,
                                     presolve_info.original_dual_size)
    return primal, dual


def row_permute_in_place(matrix, old_row_to_new):
    """Rows of a CSC matrix moved by the map ``old_row_to_new`` (0-based) without
    building a second matrix (preprocess.jl:598-622): the index and value arrays
    are rewritten column by column, rows ascending inside each column.  Not
    checked: that the map is a permutation."""
    assert sp.isspmatrix_csc(matrix)
    old_row_to_new = np.asarray(old_row_to_new)
    new_rows = old_row_to_new[matrix.indices]
    column_of = np.repeat(np.arange(matrix.shape[1]), np.diff(matrix.indptr))
    order = np.lexsort((new_rows, column_of))      # stable: by column, then by new row
    matrix.indices[:] = new_rows[order]
    matrix.data[:] = matrix.data[order]
    matrix.has_sorted_indices = True
