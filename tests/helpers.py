"""Shared test helpers: the reference's test LPs
(test/shared_test_qp_problems.jl) restated as data, and glue between the
product's problem type and the CPU oracle."""
import numpy as np
import scipy.sparse as sp

from firstorderlp_jl_amd import (QuadraticProgrammingProblem,
                                 linear_programming_problem)
from oracle.oracle import OracleState

INF = np.inf


def example_lp():
    """test/shared_test_qp_problems.jl:30-44; optimum x=[1,0,6,2], y=[.5,4,0]."""
    return linear_programming_problem(
        [0.0, 0.0, 0.0, 0.0], [2.0, 4.0, 6.0, 3.0], [5.0, 2.0, 1.0, 1.0], -14.0,
        [[2.0, 1.0, 1.0, 2.0], [1.0, 0.0, 1.0, 0.0], [0.0, 0.0, 1.0, -1.0]],
        [12.0, 7.0, 1.0], 1)


def example_lp_without_bounds():
    """shared_test_qp_problems.jl:55-65; optimum x=[2], y=[1]."""
    return linear_programming_problem([-INF], [INF], [-1.0], 0.0, [[-1.0]],
                                      [-2.0], 0)


def example_qp():
    """shared_test_qp_problems.jl:79-93; optimum x=[.2,.8], y=[.2]."""
    return QuadraticProgrammingProblem(
        [0.0, 0.0], [1.0, 1.0], [[4.0, 0.0], [0.0, 1.0]], [-1.0, -1.0], -0.0,
        [[-1.0, -1.0]], [-1.0], 0)


def example_qp2():
    """shared_test_qp_problems.jl:107-121; optimum x=[.25,0], y=[0]."""
    return QuadraticProgrammingProblem(
        [0.0, 0.0], [1.0, 1.0], [[4.0, 0.0], [0.0, 1.0]], [-1.0, 1.0], -0.0,
        [[-1.0, -1.0]], [-1.0], 0)


def example_cc_lp():
    """shared_test_qp_problems.jl:139-153."""
    return linear_programming_problem(
        [0.0] * 6, [1.0] * 6, [-1.0, -1.0, 1.0, -1.0, 1.0, -1.0], 4.0,
        [[0.0, -1.0, 1.0, 0.0, 0.0, -1.0], [0.0, 0.0, 0.0, -1.0, 1.0, -1.0],
         [-1.0, -1.0, 0.0, 1.0, 0.0, 0.0]], [-1.0, -1.0, -1.0], 0)


def example_cc_star_lp():
    """shared_test_qp_problems.jl:160-174."""
    return linear_programming_problem(
        [0.0] * 6, [1.0] * 6, [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], 3.0,
        [[-1.0, -1.0, 0.0, 1.0, 0.0, 0.0], [-1.0, 0.0, -1.0, 0.0, 1.0, 0.0],
         [0.0, -1.0, -1.0, 0.0, 0.0, 1.0]], [-1.0, -1.0, -1.0], 0)


def example_lp_dependent_rows():
    """shared_test_qp_problems.jl:192-206."""
    return linear_programming_problem(
        [0.0] * 4, [INF] * 4, [1.0, 2.0, 3.0, 4.0], 0.0,
        [[1.0, 1.0, 1.0, 1.0], [1.0, 1.0, 1.0, 1.0], [1.0, 0.0, 0.0, 1.0]],
        [2.0, 2.0, 1.0], 3)


def oracle_from_problem(p):
    A, Q = p.constraint_matrix, p.objective_matrix
    return OracleState(A.shape[0], A.shape[1], A.indptr, A.indices, A.data,
                       p.objective_vector, p.right_hand_side,
                       p.variable_lower_bound, p.variable_upper_bound,
                       p.num_equalities, Q.indptr, Q.indices, Q.data)


def initial_step_and_weight(p):
    """pdhg.jl:821-826 (1/norm(A, Inf) = 1/max|A_ij|) and
    select_initial_primal_weight (saddle_point.jl:1049-1075, unit norms)."""
    A = p.constraint_matrix
    step = 1.0 / np.abs(A.data).max()
    cn = np.sqrt(np.sum(p.objective_vector ** 2))
    bn = np.sqrt(np.sum(p.right_hand_side ** 2))
    pw = cn / bn if cn > 0 and bn > 0 else 1.0
    return step, pw


def skewed_lp(m, n, seed, dense_rows=1, dense_cols=1, base_nnz=4):
    """Random LP with a few very long rows/columns (exercises the long-row
    split path, like the PageRank LP's dense equality row and the L1-SVM
    intercept column)."""
    rng = np.random.default_rng(seed)
    A = sp.random(m, n, density=min(1.0, base_nnz / n), format="lil",
                  random_state=np.random.RandomState(seed),
                  data_rvs=rng.standard_normal)
    for r in range(dense_rows):
        A[r, :] = rng.standard_normal(n)
    for c in range(dense_cols):
        A[:, n - 1 - c] = rng.standard_normal((m, 1))
    A = A.tocsc()
    x0 = rng.random(n)
    num_eq = m // 3
    b = A @ x0
    b[num_eq:] -= rng.random(m - num_eq)
    y0 = rng.standard_normal(m)
    y0[num_eq:] = np.abs(y0[num_eq:])
    c = A.T @ y0 + rng.random(n) * (rng.random(n) < 0.5)
    lb = np.where(rng.random(n) < 0.2, -INF, 0.0)
    ub = np.where(rng.random(n) < 0.5, INF, 2.0)
    return linear_programming_problem(lb, ub, c, 0.0, A, b, num_eq)
