"""The reference's PDHG known-answer tests (test/test_primal_dual_hybrid_gradient.jl)
restated once and run against two engines: the CPU oracle (pins the oracle +
host driver, ``-m "not gpu"``) and the HIP engine through the C ABI (``-m gpu``)."""
import numpy as np

from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (
    AdaptiveStepsizeParams, ConstantStepsizeParams,
    MalitskyPockStepsizeParameters, PdhgParameters, optimize)
from firstorderlp_jl_amd.saddle_point import (RestartScheme,
                                              RestartToCurrentMetric,
                                              construct_restart_parameters)
from firstorderlp_jl_amd.solve_log import RestartChoice, TerminationReason
from firstorderlp_jl_amd.termination import (L_INF,
                                             construct_termination_criteria)
from tests import helpers as H


def terminate_on_iteration_limit(n):
    """test/utilities.jl:85-97"""
    return construct_termination_criteria(
        optimality_norm=L_INF, eps_optimal_absolute=0.0, eps_optimal_relative=0.0,
        eps_primal_infeasible=0.0, eps_dual_infeasible=0.0, time_sec_limit=100.0,
        iteration_limit=n, kkt_matrix_pass_limit=np.inf)


def generate_primal_dual_hybrid_gradient_params(
        l_inf_ruiz_iterations=0, l2_norm_rescaling=False,
        pock_chambolle_alpha=None, iteration_limit=200, primal_importance=1.0,
        scale_invariant_initial_primal_weight=True, verbosity=0,
        record_iteration_stats=True, restart_scheme=RestartScheme.NO_RESTARTS,
        restart_frequency_if_fixed=100, artificial_restart_threshold=0.5,
        sufficient_reduction_for_restart=0.1, necessary_reduction_for_restart=0.8,
        primal_weight_update_smoothing=0.5, termination_evaluation_frequency=5,
        use_approximate_localized_duality_gap=False,
        restart_to_current_metric=RestartToCurrentMetric.GAP_OVER_DISTANCE_SQUARED,
        step_size_policy="adaptive"):
    """test/test_primal_dual_hybrid_gradient.jl:15-74"""
    if step_size_policy == "malitsky-pock":
        policy = MalitskyPockStepsizeParameters(0.7, 0.99, 1.0)
    elif step_size_policy == "constant":
        policy = ConstantStepsizeParams()
    else:
        policy = AdaptiveStepsizeParams(0.3, 0.6)
    restart_params = construct_restart_parameters(
        restart_scheme, restart_to_current_metric, restart_frequency_if_fixed,
        artificial_restart_threshold, sufficient_reduction_for_restart,
        necessary_reduction_for_restart, primal_weight_update_smoothing,
        use_approximate_localized_duality_gap)
    return PdhgParameters(l_inf_ruiz_iterations, l2_norm_rescaling,
                          pock_chambolle_alpha, primal_importance,
                          scale_invariant_initial_primal_weight, verbosity,
                          record_iteration_stats, termination_evaluation_frequency,
                          terminate_on_iteration_limit(iteration_limit),
                          restart_params, policy)


def _close(a, b, atol):
    np.testing.assert_allclose(a, b, rtol=0, atol=atol)


X_LP, Y_LP = [1.0, 0.0, 6.0, 2.0], [0.5, 4.0, 0.0]


def _any_restart_to_average(output):
    return any(it.restart_used == RestartChoice.RESTART_CHOICE_RESTART_TO_AVERAGE
               for it in output.iteration_stats)


# Each case: name -> callable(factory); file:line of the reference testset.
def low_precision(f):                              # :77-87
    out = optimize(generate_primal_dual_hybrid_gradient_params(iteration_limit=300), H.example_lp(), f)
    _close(out.primal_solution, X_LP, 1e-4)
    _close(out.dual_solution, Y_LP, 1e-4)
    assert out.iteration_count == 300
    assert out.termination_reason == TerminationReason.TERMINATION_REASON_ITERATION_LIMIT


def terminate_with_optimal_solution(f):            # :88-98
    params = generate_primal_dual_hybrid_gradient_params(iteration_limit=1000)
    params.termination_criteria.eps_optimal_absolute = 1e-8
    out = optimize(params, H.example_lp(), f)
    assert out.termination_reason == TerminationReason.TERMINATION_REASON_OPTIMAL


def fixed_frequency_restart(f):                    # :116-129
    out = optimize(generate_primal_dual_hybrid_gradient_params(
        iteration_limit=500, restart_scheme=RestartScheme.FIXED_FREQUENCY,
        restart_frequency_if_fixed=30), H.example_lp(), f)
    _close(out.primal_solution, X_LP, 1e-9)
    _close(out.dual_solution, Y_LP, 1e-9)


def adaptive_restart_heuristic(f):                 # :130-147
    out = optimize(generate_primal_dual_hybrid_gradient_params(
        iteration_limit=600, restart_scheme=RestartScheme.ADAPTIVE_NORMALIZED), H.example_lp(), f)
    _close(out.primal_solution, X_LP, 1e-9)
    _close(out.dual_solution, Y_LP, 1e-9)
    assert _any_restart_to_average(out)


def constant_step_no_smoothing(f):                 # :149-172
    out = optimize(generate_primal_dual_hybrid_gradient_params(
        iteration_limit=700, primal_weight_update_smoothing=0.0,
        restart_scheme=RestartScheme.ADAPTIVE_NORMALIZED, step_size_policy="constant"),
        H.example_lp(), f)
    _close(out.primal_solution, X_LP, 1e-9)
    _close(out.dual_solution, Y_LP, 1e-9)
    assert _any_restart_to_average(out)
    step = out.iteration_stats[0].step_size
    assert all(s.step_size == step for s in out.iteration_stats)


def no_restart_to_current(f):                      # :174-192
    out = optimize(generate_primal_dual_hybrid_gradient_params(
        iteration_limit=600, restart_scheme=RestartScheme.ADAPTIVE_NORMALIZED,
        restart_to_current_metric=RestartToCurrentMetric.NO_RESTART_TO_CURRENT), H.example_lp(), f)
    _close(out.primal_solution, X_LP, 1e-9)
    _close(out.dual_solution, Y_LP, 1e-9)
    assert _any_restart_to_average(out)


def gap_over_distance(f):                          # :194-212
    out = optimize(generate_primal_dual_hybrid_gradient_params(
        iteration_limit=600, restart_scheme=RestartScheme.ADAPTIVE_NORMALIZED,
        restart_to_current_metric=RestartToCurrentMetric.GAP_OVER_DISTANCE), H.example_lp(), f)
    _close(out.primal_solution, X_LP, 1e-9)
    _close(out.dual_solution, Y_LP, 1e-9)
    assert _any_restart_to_average(out)


def adaptive_restart_zero_objective(f):            # :214-227
    params = generate_primal_dual_hybrid_gradient_params(
        iteration_limit=200, restart_scheme=RestartScheme.ADAPTIVE_NORMALIZED)
    p = H.example_lp()
    p.objective_vector = np.zeros(4)
    params.termination_criteria.eps_optimal_absolute = 1e-8
    out = optimize(params, p, f)
    assert out.termination_reason == TerminationReason.TERMINATION_REASON_OPTIMAL


def approximate_localized_duality_gap(f):          # :229-243
    params = generate_primal_dual_hybrid_gradient_params(
        iteration_limit=300, restart_scheme=RestartScheme.ADAPTIVE_NORMALIZED,
        use_approximate_localized_duality_gap=True)
    p = H.example_lp()
    p.objective_vector = np.zeros(4)
    params.termination_criteria.eps_optimal_absolute = 1e-8
    out = optimize(params, p, f)
    assert out.termination_reason == TerminationReason.TERMINATION_REASON_OPTIMAL


def malitsky_pock_no_smoothing(f):                 # :245-259
    out = optimize(generate_primal_dual_hybrid_gradient_params(
        iteration_limit=700, primal_weight_update_smoothing=0.0,
        restart_scheme=RestartScheme.ADAPTIVE_NORMALIZED, step_size_policy="malitsky-pock"),
        H.example_lp(), f)
    _close(out.primal_solution, X_LP, 1e-9)
    _close(out.dual_solution, Y_LP, 1e-9)


def malitsky_pock_smoothing(f):                    # :261-274
    out = optimize(generate_primal_dual_hybrid_gradient_params(
        iteration_limit=700, restart_scheme=RestartScheme.ADAPTIVE_NORMALIZED,
        step_size_policy="malitsky-pock"), H.example_lp(), f)
    _close(out.primal_solution, X_LP, 1e-9)
    _close(out.dual_solution, Y_LP, 1e-9)


def quadratic_programming_1(f):                    # :276-286
    out = optimize(generate_primal_dual_hybrid_gradient_params(iteration_limit=200), H.example_qp(), f)
    _close(out.primal_solution, [0.2, 0.8], 1e-4)
    _close(out.dual_solution, [0.2], 1e-4)


def quadratic_programming_2(f):                    # :287-297
    out = optimize(generate_primal_dual_hybrid_gradient_params(iteration_limit=200), H.example_qp2(), f)
    _close(out.primal_solution, [0.25, 0.0], 1e-4)
    _close(out.dual_solution, [0.0], 1e-4)


def l2_norm_rescaling(f):                          # :299-310
    out = optimize(generate_primal_dual_hybrid_gradient_params(
        l2_norm_rescaling=True, iteration_limit=200), H.example_qp2(), f)
    _close(out.primal_solution, [0.25, 0.0], 1e-4)
    _close(out.dual_solution, [0.0], 1e-4)


def ruiz(f):                                       # :311-322
    out = optimize(generate_primal_dual_hybrid_gradient_params(
        l_inf_ruiz_iterations=10, iteration_limit=200), H.example_qp2(), f)
    _close(out.primal_solution, [0.25, 0.0], 1e-4)
    _close(out.dual_solution, [0.0], 1e-4)


def pock_chambolle(f):                             # :323-335
    out = optimize(generate_primal_dual_hybrid_gradient_params(
        pock_chambolle_alpha=1.0, iteration_limit=3000), H.example_lp(), f)
    _close(out.primal_solution, X_LP, 1e-4)
    _close(out.dual_solution, Y_LP, 1e-4)


def high_precision(f):                             # :337-347
    out = optimize(generate_primal_dual_hybrid_gradient_params(iteration_limit=800), H.example_lp(), f)
    _close(out.primal_solution, X_LP, 1e-9)
    _close(out.dual_solution, Y_LP, 1e-9)


def infeasible_instance(f):                        # :348-360
    p = H.example_lp()
    p.right_hand_side[2] = 8
    out = optimize(generate_primal_dual_hybrid_gradient_params(iteration_limit=800), p, f)
    assert out.termination_reason == TerminationReason.TERMINATION_REASON_PRIMAL_INFEASIBLE


def lp_without_bounds(f):                          # :361-371
    out = optimize(generate_primal_dual_hybrid_gradient_params(iteration_limit=400),
                   H.example_lp_without_bounds(), f)
    _close(out.primal_solution, [2.0], 1e-9)
    _close(out.dual_solution, [1.0], 1e-9)


def _check_cc(out):
    tol = 1e-14
    _close(out.primal_solution, [1.0, 1.0, 0.0, 1.0, 0.0, 0.0], tol)
    final_stats = out.iteration_stats[-1]
    assert abs(final_stats.convergence_information[0].dual_objective - 1.0) <= tol
    assert np.all(out.dual_solution >= 0.0)
    assert out.dual_solution[0] + out.dual_solution[1] >= 1.0 - tol


def correlation_clustering_triangle_plus(f):       # :372-390
    out = optimize(generate_primal_dual_hybrid_gradient_params(iteration_limit=15), H.example_cc_lp(), f)
    _check_cc(out)


def numerical_error(f):                            # :391-412
    out = optimize(generate_primal_dual_hybrid_gradient_params(iteration_limit=150), H.example_cc_lp(), f)
    assert out.termination_reason == TerminationReason.TERMINATION_REASON_NUMERICAL_ERROR
    _check_cc(out)


def correlation_clustering_star(f):                # :413-423
    out = optimize(generate_primal_dual_hybrid_gradient_params(iteration_limit=100), H.example_cc_star_lp(), f)
    _close(out.primal_solution, [0.5, 0.5, 0.5, 0.0, 0.0, 0.0], 1e-6)
    _close(out.dual_solution, [0.5, 0.5, 0.5], 1e-6)


CASES = [low_precision, terminate_with_optimal_solution, fixed_frequency_restart,
         adaptive_restart_heuristic, constant_step_no_smoothing,
         no_restart_to_current, gap_over_distance,
         adaptive_restart_zero_objective, approximate_localized_duality_gap,
         malitsky_pock_no_smoothing, malitsky_pock_smoothing,
         quadratic_programming_1, quadratic_programming_2, l2_norm_rescaling,
         ruiz, pock_chambolle, high_precision, infeasible_instance,
         lp_without_bounds, correlation_clustering_triangle_plus,
         numerical_error, correlation_clustering_star]
