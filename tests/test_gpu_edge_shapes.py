"""Degenerate problem shapes through the C ABI: no constraints at all (m = 0),
an all-zero constraint matrix, a 1x1 problem, empty rows and empty columns.
The reference accepts all of them (SparseMatrixCSC with empty columns / rows);
the device layouts must not launch empty grids or index past empty arrays.
Trial vectors are compared with the oracle to 1e-12 (the three step scalars are
reduced in a different order, so the adaptive step may differ in the last bits)."""
import numpy as np
import pytest
import scipy.sparse as sp

from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (
    AdaptiveStepsizeParams, PdhgSolverState, take_step)
from firstorderlp_jl_amd.quadratic_programming import linear_programming_problem
from tests.oracle_engine import OracleEngine

pytestmark = pytest.mark.gpu


def _lp(m, n, A, num_eq):
    return linear_programming_problem(
        variable_lower_bound=np.zeros(n), variable_upper_bound=np.full(n, 2.0),
        objective_vector=np.arange(1, n + 1, dtype=float) * (-1.0) ** np.arange(n),
        objective_constant=0.0, constraint_matrix=sp.csc_matrix(A, shape=(m, n)),
        right_hand_side=np.ones(m), num_equalities=num_eq)


CASES = {
    "no_constraints": lambda: _lp(0, 3, sp.csc_matrix((0, 3)), 0),
    "all_zero_matrix": lambda: _lp(3, 3, sp.csc_matrix((3, 3)), 1),
    "one_by_one": lambda: _lp(1, 1, np.array([[2.0]]), 1),
    "empty_rows_and_columns": lambda: _lp(4, 5, np.array([[0, 0, 0, 0, 0], [1, 0, 2, 0, 0],
                                                          [0, 0, 0, 0, 0], [0, 0, 3, 0, 4.0]]), 1),
    "single_column": lambda: _lp(5, 1, np.array([[1.0], [0.0], [-2.0], [0.0], [3.0]]), 2),
    "single_row": lambda: _lp(1, 6, np.array([[1.0, 0, -2.0, 0, 0, 3.0]]), 0),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_degenerate_shapes_match_oracle(gpu_required, name):
    p = CASES[name]()
    g, o = HipPdhgEngine.from_problem(p), OracleEngine.from_problem(p)
    sg = PdhgSolverState(g, step_size=0.3, primal_weight=1.0)
    so = PdhgSolverState(o, step_size=0.3, primal_weight=1.0)
    for _ in range(12):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), sg)
        take_step(AdaptiveStepsizeParams(0.3, 0.6), so)
        assert sg.numerical_error == so.numerical_error
        if sg.numerical_error:
            break
    assert sg.total_number_iterations == so.total_number_iterations
    for a, b in zip(g.get_current(), o.get_current()):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)
    for a, b in zip(g.get_average(), o.get_average()):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)
    m, n = p.constraint_matrix.shape
    x, y = np.arange(1.0, n + 1), np.arange(1.0, m + 1)
    assert np.array_equal(g.spmv(x), p.constraint_matrix @ x)
    assert np.array_equal(g.spmv_t(y), p.constraint_matrix.T @ y)


@pytest.mark.parametrize("shards", [2, 3, 5])
@pytest.mark.parametrize("name", sorted(CASES))
def test_degenerate_shapes_on_shard_groups(gpu_required, name, shards):
    """The same shapes through row-shard groups: ranks without rows, ranks whose owned column
    slice is empty (n < world x 16), matrices without nonzeros -- plus the device evaluation
    and a restart on them."""
    p = CASES[name]()
    g, o = HipPdhgEngine.from_problem(p, device_ids=[0] * shards), OracleEngine.from_problem(p)
    assert g.dist_info()["world"] == shards
    sg = PdhgSolverState(g, step_size=0.3, primal_weight=1.0)
    so = PdhgSolverState(o, step_size=0.3, primal_weight=1.0)
    for _ in range(12):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), sg)
        take_step(AdaptiveStepsizeParams(0.3, 0.6), so)
        assert sg.numerical_error == so.numerical_error
        if sg.numerical_error:
            break
    assert sg.total_number_iterations == so.total_number_iterations
    for a, b in zip(g.get_current(), o.get_current()):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)
    for a, b in zip(g.get_average(), o.get_average()):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)
    m, n = p.constraint_matrix.shape
    x, y = np.arange(1.0, n + 1), np.arange(1.0, m + 1)
    assert np.array_equal(g.spmv(x), p.constraint_matrix @ x)
    np.testing.assert_allclose(g.spmv_t(y), p.constraint_matrix.T @ y, rtol=1e-14, atol=1e-14)
    # evaluation branch and a restart on the degenerate group
    g.set_original_problem(np.ones(m), np.ones(n), p.objective_vector, p.right_hand_side,
                           p.variable_lower_bound, p.variable_upper_bound)
    single = HipPdhgEngine.from_problem(p)
    single.set_original_problem(np.ones(m), np.ones(n), p.objective_vector, p.right_hand_side,
                                p.variable_lower_bound, p.variable_upper_bound)
    xg, yg = g.get_current()
    single.set_current(xg, yg)
    np.testing.assert_allclose(g.eval_point(0), single.eval_point(0), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(g.trust_region_bound(0, 1.0, 1.0, 0.5, 0)[:6], single.trust_region_bound(0, 1.0, 1.0, 0.5, 0)[:6],
                               rtol=1e-10, atol=1e-12)
    if not sg.numerical_error and g.average_info()[0] > 0:
        g.restart_to_average()
        xa, ya = g.get_current()
        np.testing.assert_allclose(g.get_dual_product(), p.constraint_matrix.T @ ya, rtol=1e-13, atol=1e-13)
