"""The reference's bound_optimal_objective known answers
(test/test_trust_region_utils.jl:212-327) through the DEVICE implementation
(pdhg_trust_region_bound)."""
import numpy as np
import pytest

from firstorderlp_jl_amd import HipPdhgEngine, _lib
from firstorderlp_jl_amd.evaluation import DeviceEvaluator, POINT_CURRENT
from firstorderlp_jl_amd.quadratic_programming import ScaledQpProblem
from firstorderlp_jl_amd.termination import cached_quadratic_program_info
from tests import trust_region_cases as C

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("small_eval", ["1", "0"], ids=["one_workgroup_kernel", "pass_by_pass"])
@pytest.mark.parametrize("case", C.CASES, ids=lambda c: c[0])
def test_bound_optimal_objective_device(gpu_required, monkeypatch, case, small_eval):
    monkeypatch.setenv("PDHG_SMALL_EVAL", small_eval)
    name, maker, x, y, radius, norm, expected = case
    p = maker()
    eng = HipPdhgEngine.from_problem(p)
    sp_ = ScaledQpProblem(p, p, np.ones(p.num_constraints), np.ones(p.num_variables))
    ev = DeviceEvaluator(eng, sp_, cached_quadratic_program_info(p))
    eng.set_current(np.array(x), np.array(y))
    r = ev.bound(POINT_CURRENT, 1.0, 1.0, radius, norm, False)
    C.check(r, expected, 1e-12)


@pytest.mark.parametrize("case", __import__("tests.stats_cases", fromlist=["CASES"]).CASES, ids=lambda c: c[0])
def test_convergence_information_device(gpu_required, case):
    """test/test_iteration_stats.jl:118-308: the ConvergenceInformation part through
    pdhg_eval_point (the device path evaluates the rays at the iterate itself, as
    optimize does, so the InfeasibilityInformation of those cases is host-only)."""
    from tests import stats_cases as S
    name, lp, x, y, xr, yr, want_ci, want_ii = case
    eng = HipPdhgEngine.from_problem(lp)
    sp_ = ScaledQpProblem(lp, lp, np.ones(lp.num_constraints), np.ones(lp.num_variables))
    ev = DeviceEvaluator(eng, sp_, cached_quadratic_program_info(lp))
    eng.set_current(np.array(x), np.array(y))
    from firstorderlp_jl_amd.termination import construct_termination_criteria
    from firstorderlp_jl_amd.solve_log import PointType
    tc = construct_termination_criteria(eps_optimal_absolute=1e-6, eps_optimal_relative=1e-6)
    st = ev.iteration_stats(POINT_CURRENT, tc, True, 6, 5.0, 1.5, 1.0, 1.0, PointType.POINT_TYPE_CURRENT_ITERATE)
    S.check_ci(st.convergence_information[0], want_ci, 1e-14)
    assert st.iteration_number == 5
