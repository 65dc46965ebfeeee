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


def sparse_uniform(m, n, density, seed, fmt="csc"):
    """A random sparse matrix with uniform(0, 1) entries like sp.random's default, its cells drawn WITH replacement
    (repeats summed): sp.random draws without replacement from the m * n cells, seconds at 12 000 x 12 000."""
    rng = np.random.default_rng(seed)
    k = int(round(density * m * n))
    A = sp.coo_matrix((rng.random(k), (rng.integers(0, m, k), rng.integers(0, n, k))), shape=(m, n)).asformat(fmt)
    A.sum_duplicates()
    A.sort_indices()
    return A


def skewed_lp(m, n, seed, dense_rows=1, dense_cols=1, base_nnz=4):
    """Random LP with a few very long rows/columns (exercises the long-row
    split path, like the PageRank LP's dense equality row and the L1-SVM
    intercept column)."""
    rng = np.random.default_rng(seed)
    # the sparse part as triplets drawn with replacement (sp.random draws WITHOUT replacement from the m * n cells: 40-100 s
    # at 30 000 x 40 000; the few repeated cells are summed by the constructor below)
    k = int(round(min(1.0, base_nnz / n) * m * n))
    br, bc, bv = rng.integers(0, m, k), rng.integers(0, n, k), rng.standard_normal(k)
    # the dense rows 0 .. dense_rows - 1 and the dense columns n - 1, n - 2, ... replace what the sparse part holds there
    keep = (br >= dense_rows) & (bc < n - dense_cols)
    rows, cols, vals = [br[keep]], [bc[keep]], [bv[keep]]
    for r in range(dense_rows):
        rows.append(np.full(n, r)); cols.append(np.arange(n)); vals.append(rng.standard_normal(n))
    for c in range(dense_cols):
        rows.append(np.arange(dense_rows, m)); cols.append(np.full(m - dense_rows, n - 1 - c)); vals.append(rng.standard_normal(m - dense_rows))
    A = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(m, n))
    A.sort_indices()
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


def bitexact_row_limit():
    """Rows of at most this many entries are bit-identical to the oracle's sequential sums in the row order the test runs
    in (conftest.row_order_mode sets PDHG_ROW_ORDER): 2048 (BLOCK_NNZ) in strict order, 256 (RELAXED_MIN_ROW) in the
    shipped relaxed order; longer rows are within 1e-13 * sum |a x| of them (spmv_kernels.hpp)."""
    import os
    return 256 if os.environ.get("PDHG_ROW_ORDER", "relaxed") == "relaxed" else 2048


def assert_rows_match_oracle(got, want, row_nnz, abs_scale, label=""):
    """got / want: a product's rows on the device / from the oracle; row_nnz: entries per row; abs_scale: sum |a x| per
    row (or any bound of it).  Short rows bitwise, long rows within 1e-13 of the scale."""
    short = row_nnz <= bitexact_row_limit()
    assert np.array_equal(got[short], want[short]), label + ": short rows differ"
    if np.any(~short):
        assert np.all(np.abs(got[~short] - want[~short]) <= 1e-13 * abs_scale[~short] + 1e-300), label + ": long rows beyond 1e-13 * sum|a x|"


def assert_products_match_oracle(eng, A, x, y, forced_sweep=False, label=""):
    """A x and A'y of `eng` against the oracle's sequential loops in the row order the test runs in: rows up to the
    bit-exact limit bitwise, longer rows within 1e-13 * sum |a x| (and never worse than that anywhere).
    forced_sweep: PDHG_SPMV=tiled put the sweep on a matrix whose rows have long runs inside one tile (the builder itself
    would stream it).  In relaxed order a CHUNK holding a same-row run of more than 8 entries is tree-reduced as a whole
    (tiled_chunk_relaxed), so only rows of at most 8 entries are then guaranteed bitwise; strict order is unaffected."""
    import os
    from oracle import oracle as orc
    import scipy.sparse as sp
    A = sp.csc_matrix(A)
    m, n = A.shape
    relaxed = os.environ.get("PDHG_ROW_ORDER", "relaxed") == "relaxed"
    limit = (8 if forced_sweep else 256) if relaxed else 2048
    absA = abs(A).tocsr()
    for got, want, nnz_per, scale, name in (
            (eng.spmv(x), orc.spmv(m, n, A.indptr, A.indices, A.data, x), np.diff(A.tocsr().indptr), absA @ np.abs(x), "A x"),
            (eng.spmv_t(y), orc.spmv_t(m, n, A.indptr, A.indices, A.data, y), np.diff(A.indptr), absA.T @ np.abs(y), "A'y")):
        short = nnz_per <= limit
        assert np.array_equal(got[short], want[short]), f"{label} {name}: rows of <= {limit} entries differ from the oracle"
        assert np.all(np.abs(got - want) <= 1e-13 * scale + 1e-300), f"{label} {name}: beyond 1e-13 * sum |a x|"
