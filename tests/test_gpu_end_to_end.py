"""optimize() end to end with the reference's solve_qp.jl defaults: everything on
the device (HIP engine: device rescaling, device evaluation/restarts/trust
region) against everything on the host (CPU oracle engine: host rescaling, numpy
evaluation).  The two runs take reduction sums in different orders, so restart
decisions may fork; what must agree is the outcome: OPTIMAL, objectives within
the termination tolerance, primal solutions close, iteration counts comparable."""
import numpy as np
import pytest

from firstorderlp_jl_amd.generators import pagerank_lp, random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (AdaptiveStepsizeParams, PdhgParameters,
                                                             optimize)
from firstorderlp_jl_amd.saddle_point import (RestartScheme, RestartToCurrentMetric,
                                              construct_restart_parameters)
from firstorderlp_jl_amd.termination import construct_termination_criteria
from tests.oracle_engine import OracleEngine

pytestmark = pytest.mark.gpu


def _params(tol, limit):
    tc = construct_termination_criteria(eps_optimal_absolute=tol, eps_optimal_relative=tol,
                                        iteration_limit=limit)
    rp = construct_restart_parameters(RestartScheme.ADAPTIVE_NORMALIZED,
                                      RestartToCurrentMetric.GAP_OVER_DISTANCE_SQUARED,
                                      1000, 0.5, 0.1, 0.9, 0.5, False)
    return PdhgParameters(10, False, 1.0, 1.0, True, 0, True, 40, tc, rp, AdaptiveStepsizeParams(0.3, 0.6))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name,maker", [("random", lambda: random_lp(12000, 10000, 8, seed=42)),
                                        ("pagerank", lambda: pagerank_lp(20000, seed=3))])
def test_device_solve_matches_host_solve(gpu_required, name, maker):
    tol = 1e-6
    dev = optimize(_params(tol, 40000), maker())
    host = optimize(_params(tol, 40000), maker(), OracleEngine.from_problem)
    assert dev.termination_string == host.termination_string == "OPTIMAL"
    cd = dev.iteration_stats[-1].convergence_information[0]
    ch = host.iteration_stats[-1].convergence_information[0]
    scale = 1.0 + abs(ch.primal_objective)
    assert abs(cd.primal_objective - ch.primal_objective) <= 50 * tol * scale
    assert abs(cd.dual_objective - ch.dual_objective) <= 50 * tol * scale
    assert max(cd.relative_l2_primal_residual, cd.relative_l2_dual_residual, cd.relative_optimality_gap) <= tol
    ratio = dev.iteration_count / host.iteration_count
    assert 0.5 <= ratio <= 2.0, (dev.iteration_count, host.iteration_count)
    # both are 1e-6-optimal points of the same LP (which need not have a unique solution)
    diff = np.linalg.norm(dev.primal_solution - host.primal_solution)
    assert diff <= 0.05 * (1.0 + np.linalg.norm(host.primal_solution))


def test_dropped_checks_skip_the_bound_estimates_and_kept_stats_are_unchanged(gpu_required):
    """update_objective_bound_estimates (pdhg.jl:938-945) only feeds stats that are KEPT (the log, the final report):
    with record_iteration_stats = false the checks that do not terminate skip its two trust-region problems.  The solve
    itself must not notice -- same iterations, same solution, bit for bit -- and the final (kept) stats carry the same
    three entries as when every check records."""
    import dataclasses
    p = random_lp(12000, 10000, 8, seed=42)
    keep = _params(1e-5, 20000)
    drop = dataclasses.replace(keep, record_iteration_stats=False)
    a, b = optimize(keep, p), optimize(drop, p)
    assert a.termination_string == b.termination_string == "OPTIMAL"
    assert a.iteration_count == b.iteration_count
    assert np.array_equal(a.primal_solution, b.primal_solution) and np.array_equal(a.dual_solution, b.dual_solution)
    assert len(a.iteration_stats) > 3 and len(b.iteration_stats) == 1
    ma, mb = a.iteration_stats[-1].method_specific_stats, b.iteration_stats[-1].method_specific_stats
    for key in ("lagrangian_value", "estimated_lower_bound", "estimated_upper_bound"):
        assert ma[key] == mb[key], key
