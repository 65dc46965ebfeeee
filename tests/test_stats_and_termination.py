"""test/test_iteration_stats.jl and test/test_termination.jl restated for the
host implementations.  CPU only."""
import copy

import numpy as np
import pytest

from firstorderlp_jl_amd import linear_programming_problem
from firstorderlp_jl_amd import termination as T
from firstorderlp_jl_amd.iteration_stats_utils import (compute_dual_stats,
                                                       compute_iteration_stats,
                                                       max_primal_violation, primal_obj)
from firstorderlp_jl_amd.solve_log import (ConvergenceInformation, InfeasibilityInformation,
                                           IterationStats, PointType, TerminationReason)
from tests import helpers as H
from tests import stats_cases as S

INF = np.inf


def test_max_primal_violation():                     # test_iteration_stats.jl:17-35
    lp = linear_programming_problem([-1.0, -INF, -INF], [1.0, INF, INF], np.zeros(3), 0.0,
                                    [[0.0, 1.0, 0.0], [0.0, 0.0, 1.0]], [10.0, 11.0], 1)
    for x, want in (([0.0, 10.0, 11.0], 0.0), ([-2.0, 10.0, 11.0], 1.0), ([3.0, 10.0, 11.0], 2.0),
                    ([0.0, 11.0, 11.0], 1.0), ([0.0, 9.0, 11.0], 1.0), ([0.0, 11.0, 0.0], 11.0)):
        assert max_primal_violation(lp, np.array(x)) == want


def test_primal_obj():                               # :37-44
    qp = H.example_qp()
    for x, want in (([0.0, 0.0], 0.0), ([1.0, 1.0], 0.5), ([1.0, 0.0], 1.0), ([0.0, 1.0], -0.5),
                    ([0.0, -1.0], 1.5)):
        assert primal_obj(qp, np.array(x)) == want


def test_dual_stats():                               # :46-115
    lp = linear_programming_problem([-1.0, -INF], [1.0, INF], [1.0, 2.0], 0.0, [[1.0, 1.0]], [1.0], 0)
    z = np.zeros(2)
    for y, obj, res in ((0.0, -1.0, [0.0, 0.0, 2.0]), (1.0, 1.0, [0.0, 0.0, 1.0])):
        ds = compute_dual_stats(lp, z, np.array([y]))
        assert ds.dual_objective == obj and np.array_equal(ds.dual_residual, res)
    for y, obj, rinf in ((2.0, 1.0, 0.0), (3.0, 1.0, 1.0)):
        ds = compute_dual_stats(lp, z, np.array([y]))
        assert ds.dual_objective == obj and np.max(np.abs(ds.dual_residual)) == rinf
    ds = compute_dual_stats(lp, np.array([0.0, 1.0]), np.array([-1.0]))
    assert ds.dual_objective == -3.0 and np.array_equal(ds.dual_residual, [1.0, 0.0, 3.0])
    lp2 = linear_programming_problem([INF, -INF], [INF, INF], [1.0, 2.0], 0.0, [[1.0, 1.0]], [1.0], 0)
    ds = compute_dual_stats(lp2, np.array([0.0, 1.0]), np.array([-1.0]))
    assert ds.dual_objective == -1.0 and np.array_equal(ds.dual_residual, [1.0, 2.0, 3.0])
    qp = H.example_qp()
    for x, y, obj in (([0.0, 0.0], 3.0, -3.0), ([0.0, 0.0], 1.0, -1.0), ([0.5, 0.5], 1.0, -1.625)):
        ds = compute_dual_stats(qp, np.array(x), np.array([y]))
        assert ds.dual_objective == obj and np.max(np.abs(ds.dual_residual)) == 0.0


@pytest.mark.parametrize("case", S.CASES, ids=lambda c: c[0])
def test_compute_iteration_stats(case):              # :118-308
    name, lp, x, y, xr, yr, want_ci, want_ii = case
    st = compute_iteration_stats(lp, T.cached_quadratic_program_info(lp), np.array(x), np.array(y),
                                 np.array(xr), np.array(yr), 5, 1.5, 5.0, 1e-6, 1e-6, 1.0, 1.0,
                                 PointType.POINT_TYPE_CURRENT_ITERATE)
    S.check_ci(st.convergence_information[0], want_ci, 1e-15)
    ii = st.infeasibility_information[0]
    for f in S.II_FIELDS:
        assert getattr(ii, f) == want_ii.get(f, 0.0), f
    assert st.iteration_number == 5 and st.cumulative_kkt_matrix_passes == 1.5
    assert st.convergence_information[0].candidate_type == PointType.POINT_TYPE_CURRENT_ITERATE


def test_termination():                              # test_termination.jl:15-194
    none1 = InfeasibilityInformation()
    none2 = InfeasibilityInformation(primal_ray_linear_objective=-1.0, primal_ray_quadratic_norm=1.0,
                                     max_dual_ray_infeasibility=1.0)
    dual_inf = InfeasibilityInformation(primal_ray_linear_objective=-1.0)
    primal_inf = InfeasibilityInformation(dual_ray_objective=1.0)
    eps = 1e-6
    assert [T.primal_infeasibility_criteria_met(eps, i) for i in (none1, none2, dual_inf, primal_inf)] == \
        [False, False, False, True]
    assert [T.dual_infeasibility_criteria_met(eps, i) for i in (none1, none2, dual_inf, primal_inf)] == \
        [False, False, True, False]
    opt = ConvergenceInformation(primal_objective=1.0, dual_objective=1.0, l_inf_primal_variable=1.0,
                                 l2_primal_variable=1.0, l_inf_dual_variable=2.0, l2_dual_variable=2.0)
    no1 = copy.deepcopy(opt); no1.primal_objective = 10.0
    no2 = copy.deepcopy(opt); no2.l_inf_primal_residual = no2.l2_primal_residual = 1.0
    no3 = copy.deepcopy(opt); no3.l_inf_dual_residual = no3.l2_dual_residual = 1.0
    mk = lambda ci: IterationStats(iteration_number=5, cumulative_kkt_matrix_passes=100.5,
                                   cumulative_time_sec=5.0, convergence_information=[ci],
                                   infeasibility_information=[none1])
    qp_cache = T.cached_quadratic_program_info(H.example_qp())
    for norm in (T.L_INF, T.L2):
        assert [T.optimality_criteria_met(norm, 1e-4, 1e-4, c, qp_cache) for c in (no1, no2, no3, opt)] == \
            [False, False, False, True]
        tc = T.construct_termination_criteria(optimality_norm=norm, eps_optimal_absolute=1e-4,
                                              eps_optimal_relative=1e-4, eps_primal_infeasible=eps,
                                              eps_dual_infeasible=eps, time_sec_limit=100.0,
                                              iteration_limit=10, kkt_matrix_pass_limit=10000.0)
        assert T.check_termination_criteria(tc, qp_cache, mk(opt)) == TerminationReason.TERMINATION_REASON_OPTIMAL
        assert T.check_termination_criteria(tc, qp_cache, mk(no1)) is False
        tc.time_sec_limit = 1.0
        assert T.check_termination_criteria(tc, qp_cache, mk(no1)) == TerminationReason.TERMINATION_REASON_TIME_LIMIT
        tc.time_sec_limit = 10.0
        tc.iteration_limit = 1
        assert T.check_termination_criteria(tc, qp_cache, mk(no1)) == TerminationReason.TERMINATION_REASON_ITERATION_LIMIT
        tc.iteration_limit = 10
        tc.kkt_matrix_pass_limit = 40.0
        assert T.check_termination_criteria(tc, qp_cache, mk(no1)) == \
            TerminationReason.TERMINATION_REASON_KKT_MATRIX_PASS_LIMIT


def test_print_to_screen_this_iteration():           # test_iteration_stats.jl:310-342
    from firstorderlp_jl_amd.iteration_stats_utils import print_to_screen_this_iteration as f
    assert f(False, 1, 2, 10)
    assert f(False, 101, 5, 10)
    assert not f(False, 31, 5, 10)
    assert not f(False, 531, 5, 10)
    assert f(True, 124, 5, 10)
