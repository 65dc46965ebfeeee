"""Convergence / infeasibility statistics: host logic mirroring
src/iteration_stats_utils.jl.  Runs at evaluation cadence only (every
``termination_evaluation_frequency`` iterations, pdhg.jl:892-927), on the
ORIGINAL (unscaled) problem.

``ops`` abstracts the two mat-vecs so the evaluation branch can use device
bandwidth (``pdhg_spmv`` / ``pdhg_spmv_t``) instead of host scipy.
"""
import math
from dataclasses import dataclass

import numpy as np

from .quadratic_programming import (QuadraticProgrammingProblem,
                                    linear_programming_problem)
from .solve_log import (ConvergenceInformation, InfeasibilityInformation,
                        IterationStats, PointType)
from .termination import _norm2, _norm_inf


class HostOps:
    """A*x and A'*y with scipy on the host (CSR cached for A*x)."""

    def __init__(self, problem):
        self.A_csc = problem.constraint_matrix
        self.A_csr = problem.constraint_matrix.tocsr()
        self.Q = problem.objective_matrix
        self._has_q = problem.objective_matrix.nnz > 0

    def Ax(self, x):
        return self.A_csr @ x

    def ATy(self, y):
        return self.A_csc.T @ y

    def Qx(self, x):
        if not self._has_q:
            return np.zeros(self.Q.shape[0])
        return self.Q @ x


def _ops(problem, ops):
    return ops if ops is not None else HostOps(problem)


def compute_primal_residual(problem, primal_vec, ops=None):
    """iteration_stats_utils.jl:30-63"""
    ops = _ops(problem, ops)
    activities = ops.Ax(primal_vec)
    ne = problem.num_equalities
    b = problem.right_hand_side
    equality_violation = b[:ne] - activities[:ne]
    inequality_violation = np.maximum(b[ne:] - activities[ne:], 0.0)
    lower_bound_violation = np.maximum(problem.variable_lower_bound - primal_vec, 0.0)
    upper_bound_violation = np.maximum(primal_vec - problem.variable_upper_bound, 0.0)
    return np.concatenate([equality_violation, inequality_violation,
                           lower_bound_violation, upper_bound_violation])


def max_primal_violation(problem, primal_vec, ops=None):
    """iteration_stats_utils.jl:15-20"""
    return _norm_inf(compute_primal_residual(problem, primal_vec, ops))


def primal_obj(problem, primal_solution, ops=None):
    """iteration_stats_utils.jl:67-74"""
    ops = _ops(problem, ops)
    return (problem.objective_constant +
            float(problem.objective_vector @ primal_solution) +
            0.5 * float(primal_solution @ ops.Qx(primal_solution)))


@dataclass
class DualStats:
    """iteration_stats_utils.jl:78-82"""
    dual_objective: float
    dual_residual: np.ndarray
    reduced_costs: np.ndarray


def reduced_costs_dual_objective_contribution(variable_lower_bound,
                                              variable_upper_bound,
                                              reduced_costs):
    """iteration_stats_utils.jl:93-116"""
    bound_value = np.where(reduced_costs > 0.0, variable_lower_bound,
                           variable_upper_bound)
    active = reduced_costs != 0.0
    if np.any(active & ~np.isfinite(bound_value)):
        return -math.inf
    return float(np.sum(bound_value[active] * reduced_costs[active]))


def compute_reduced_costs_from_primal_gradient(variable_lower_bound,
                                               variable_upper_bound,
                                               primal_gradient):
    """iteration_stats_utils.jl:128-148"""
    bound_value = np.where(primal_gradient > 0.0, variable_lower_bound,
                           variable_upper_bound)
    return np.where(np.isfinite(bound_value), primal_gradient, 0.0)


def compute_dual_stats(problem, primal_solution, dual_solution, ops=None):
    """iteration_stats_utils.jl:157-197"""
    ops = _ops(problem, ops)
    objective_product = ops.Qx(primal_solution)
    # compute_primal_gradient (saddle_point.jl:1081-1100)
    primal_gradient = objective_product + problem.objective_vector - ops.ATy(dual_solution)
    reduced_costs = compute_reduced_costs_from_primal_gradient(
        problem.variable_lower_bound, problem.variable_upper_bound, primal_gradient)
    ne = problem.num_equalities
    dual_residual = np.maximum(-dual_solution[ne:], 0.0)
    reduced_cost_violations = primal_gradient - reduced_costs
    dual_residual = np.concatenate([dual_residual, reduced_cost_violations])
    base_dual_objective = (float(problem.right_hand_side @ dual_solution) +
                           problem.objective_constant -
                           0.5 * float(objective_product @ primal_solution))
    dual_objective = base_dual_objective + reduced_costs_dual_objective_contribution(
        problem.variable_lower_bound, problem.variable_upper_bound, reduced_costs)
    return DualStats(dual_objective, dual_residual, reduced_costs)


def corrected_dual_obj(problem, dual_stats):
    """iteration_stats_utils.jl:203-213"""
    if _norm_inf(dual_stats.dual_residual) == 0.0:
        return dual_stats.dual_objective
    return -math.inf


def _safe_div(a, b):
    """Julia float division semantics (x/0 -> Inf/NaN, no exception)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(np.float64(a) / np.float64(b))


def compute_convergence_information(problem, qp_cache, primal_iterate,
                                    dual_iterate, eps_ratio, candidate_type,
                                    ops=None):
    """iteration_stats_utils.jl:228-280"""
    ops = _ops(problem, ops)
    ci = ConvergenceInformation()
    primal_residual = compute_primal_residual(problem, primal_iterate, ops)
    ci.primal_objective = primal_obj(problem, primal_iterate, ops)
    ci.l_inf_primal_residual = _norm_inf(primal_residual)
    ci.l2_primal_residual = _norm2(primal_residual)
    ci.relative_l_inf_primal_residual = _safe_div(
        ci.l_inf_primal_residual, eps_ratio + qp_cache.l_inf_norm_primal_right_hand_side)
    ci.relative_l2_primal_residual = _safe_div(
        ci.l2_primal_residual, eps_ratio + qp_cache.l2_norm_primal_right_hand_side)
    ci.l_inf_primal_variable = _norm_inf(primal_iterate)
    ci.l2_primal_variable = _norm2(primal_iterate)

    dual_stats = compute_dual_stats(problem, primal_iterate, dual_iterate, ops)
    ci.dual_objective = dual_stats.dual_objective
    ci.l_inf_dual_residual = _norm_inf(dual_stats.dual_residual)
    ci.l2_dual_residual = _norm2(dual_stats.dual_residual)
    ci.relative_l_inf_dual_residual = _safe_div(
        ci.l_inf_dual_residual, eps_ratio + qp_cache.l_inf_norm_primal_linear_objective)
    ci.relative_l2_dual_residual = _safe_div(
        ci.l2_dual_residual, eps_ratio + qp_cache.l2_norm_primal_linear_objective)
    ci.l_inf_dual_variable = _norm_inf(dual_iterate)
    ci.l2_dual_variable = _norm2(dual_iterate)
    ci.corrected_dual_objective = corrected_dual_obj(problem, dual_stats)
    gap = abs(ci.primal_objective - ci.dual_objective)
    abs_obj = abs(ci.primal_objective) + abs(ci.dual_objective)
    ci.relative_optimality_gap = _safe_div(gap, eps_ratio + abs_obj)
    ci.candidate_type = candidate_type
    return ci


def compute_infeasibility_information(problem, primal_ray_estimate,
                                      dual_ray_estimate, candidate_type,
                                      ops=None):
    """iteration_stats_utils.jl:287-349"""
    ops = _ops(problem, ops)
    ii = InfeasibilityInformation()
    primal_ray_inf_norm = _norm_inf(primal_ray_estimate)
    if primal_ray_inf_norm != 0.0:
        primal_ray_estimate = primal_ray_estimate / primal_ray_inf_norm
    n, m = problem.num_variables, problem.num_constraints
    # the homogeneous problems share A, so they share ``ops``; they are built
    # as light views (no matrix copy)
    homogeneous_primal = _ProblemView(
        problem,
        variable_lower_bound=np.where(np.isfinite(problem.variable_lower_bound), 0.0, -np.inf),
        variable_upper_bound=np.where(np.isfinite(problem.variable_upper_bound), 0.0, np.inf),
        objective_constant=0.0, right_hand_side=np.zeros(m), zero_q=True)
    homogeneous_residual = compute_primal_residual(homogeneous_primal, primal_ray_estimate, ops)
    ii.max_primal_ray_infeasibility = _norm_inf(homogeneous_residual)
    ii.primal_ray_linear_objective = float(problem.objective_vector @ primal_ray_estimate)
    ii.primal_ray_quadratic_norm = _norm_inf(ops.Qx(primal_ray_estimate))

    homogeneous_dual = _ProblemView(problem, objective_vector=np.zeros(n),
                                    objective_constant=0.0, zero_q=True)
    hd_ops = _ZeroQOps(ops, n)
    homogeneous_dual_stats = compute_dual_stats(homogeneous_dual, primal_ray_estimate,
                                                dual_ray_estimate, hd_ops)
    scaling_factor = max(_norm_inf(dual_ray_estimate),
                         _norm_inf(homogeneous_dual_stats.reduced_costs))
    if scaling_factor != 0.0:
        ii.max_dual_ray_infeasibility = _norm_inf(homogeneous_dual_stats.dual_residual) / scaling_factor
        ii.dual_ray_objective = homogeneous_dual_stats.dual_objective / scaling_factor
    else:
        ii.max_dual_ray_infeasibility = 0.0
        ii.dual_ray_objective = 0.0
    ii.candidate_type = candidate_type
    return ii


class _ZeroQOps:
    """ops of a problem built with linear_programming_problem (Q = 0)."""

    def __init__(self, ops, n):
        self._ops, self._n = ops, n
        self.Ax, self.ATy = ops.Ax, ops.ATy

    def Qx(self, x):
        return np.zeros(self._n)


class _ProblemView:
    """A QuadraticProgrammingProblem with some vector fields replaced."""

    def __init__(self, base, zero_q=False, **overrides):
        for f in ("variable_lower_bound", "variable_upper_bound", "objective_vector",
                  "objective_constant", "right_hand_side", "num_equalities",
                  "constraint_matrix", "objective_matrix"):
            setattr(self, f, overrides.get(f, getattr(base, f)))
        self.num_variables = base.num_variables
        self.num_constraints = base.num_constraints


def compute_iteration_stats(problem, qp_cache, primal_iterate, dual_iterate,
                            primal_ray_estimate, dual_ray_estimate,
                            iteration_number, cumulative_kkt_matrix_passes,
                            cumulative_time_sec, eps_optimal_absolute,
                            eps_optimal_relative, step_size, primal_weight,
                            candidate_type, ops=None):
    """iteration_stats_utils.jl:356-407"""
    ops = _ops(problem, ops)
    stats = IterationStats()
    stats.iteration_number = int(iteration_number)
    stats.cumulative_kkt_matrix_passes = cumulative_kkt_matrix_passes
    stats.cumulative_time_sec = cumulative_time_sec
    stats.convergence_information = [compute_convergence_information(
        problem, qp_cache, primal_iterate, dual_iterate,
        _safe_div(eps_optimal_absolute, eps_optimal_relative), candidate_type, ops)]
    stats.infeasibility_information = [compute_infeasibility_information(
        problem, primal_ray_estimate, dual_ray_estimate, candidate_type, ops)]
    stats.step_size = step_size
    stats.primal_weight = primal_weight
    stats.method_specific_stats = {}
    return stats


def evaluate_unscaled_iteration_stats(scaled_problem, qp_cache,
                                      termination_criteria,
                                      record_iteration_stats, primal_solution,
                                      dual_solution, iteration, cumulative_time,
                                      cumulative_kkt_passes,
                                      eps_optimal_absolute,
                                      eps_optimal_relative, step_size,
                                      primal_weight, candidate_type,
                                      original_ops=None):
    """iteration_stats_utils.jl:413-451"""
    original_primal_solution = primal_solution / scaled_problem.variable_rescaling
    original_dual_solution = dual_solution / scaled_problem.constraint_rescaling
    return compute_iteration_stats(
        scaled_problem.original_qp, qp_cache, original_primal_solution,
        original_dual_solution, original_primal_solution, original_dual_solution,
        iteration - 1, cumulative_kkt_passes, cumulative_time,
        eps_optimal_absolute, eps_optimal_relative, step_size, primal_weight,
        candidate_type, original_ops)


def print_to_screen_this_iteration(termination_reason, iteration, verbosity,
                                   termination_evaluation_frequency):
    """Whether the stats row of this evaluation is printed
    (iteration_stats_utils.jl:459-490): always when terminating, otherwise every
    display_frequency evaluations, the frequency falling as verbosity rises."""
    if verbosity < 2:
        return False
    if termination_reason is not False:
        return True
    if verbosity >= 9:
        display_frequency = 1
    elif verbosity >= 6:
        display_frequency = 3
    elif verbosity >= 5:
        display_frequency = 10
    elif verbosity >= 4:
        display_frequency = 20
    elif verbosity >= 3:
        display_frequency = 50
    else:
        return iteration == 1
    return math.fmod((iteration - 1) / termination_evaluation_frequency, display_frequency) == 0
