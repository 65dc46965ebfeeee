"""Termination criteria: host-side scalar logic mirroring src/termination.jl."""
import enum
import math
from dataclasses import dataclass

import numpy as np

from .solve_log import TerminationReason


class OptimalityNorm(enum.Enum):
    """termination.jl:15"""
    L_INF = 0
    L2 = 1


L_INF = OptimalityNorm.L_INF
L2 = OptimalityNorm.L2

INT32_MAX = 2 ** 31 - 1


@dataclass
class TerminationCriteria:
    """termination.jl:29-98"""
    optimality_norm: OptimalityNorm = L2
    eps_optimal_absolute: float = 1.0e-6
    eps_optimal_relative: float = 1.0e-6
    eps_primal_infeasible: float = 1.0e-8
    eps_dual_infeasible: float = 1.0e-8
    time_sec_limit: float = math.inf
    iteration_limit: int = INT32_MAX
    kkt_matrix_pass_limit: float = math.inf


def construct_termination_criteria(**kw):
    """termination.jl:100-120 (same keyword names and defaults)."""
    return TerminationCriteria(**kw)


def validate_termination_criteria(criteria):
    """termination.jl:122-138"""
    if criteria.eps_primal_infeasible < 0:
        raise ValueError("eps_primal_infeasible must be nonnegative")
    if criteria.eps_dual_infeasible < 0:
        raise ValueError("eps_dual_infeasible must be nonnegative")
    if criteria.time_sec_limit <= 0:
        raise ValueError("time_sec_limit must be positive")
    if criteria.iteration_limit <= 0:
        raise ValueError("iteration_limit must be positive")
    if criteria.kkt_matrix_pass_limit <= 0:
        raise ValueError("kkt_matrix_pass_limit must be positive")


@dataclass
class CachedQuadraticProgramInfo:
    """termination.jl:144-149"""
    l_inf_norm_primal_linear_objective: float
    l_inf_norm_primal_right_hand_side: float
    l2_norm_primal_linear_objective: float
    l2_norm_primal_right_hand_side: float


def _norm_inf(v):
    return float(np.max(np.abs(v))) if len(v) else 0.0


def _norm2(v):
    return float(np.sqrt(np.sum(np.square(v)))) if len(v) else 0.0


def cached_quadratic_program_info(qp):
    """termination.jl:151-158"""
    return CachedQuadraticProgramInfo(
        _norm_inf(qp.objective_vector), _norm_inf(qp.right_hand_side),
        _norm2(qp.objective_vector), _norm2(qp.right_hand_side))


def optimality_criteria_met(optimality_norm, abs_tol, rel_tol, ci, qp_cache):
    """termination.jl:163-193"""
    abs_obj = abs(ci.primal_objective) + abs(ci.dual_objective)
    gap = abs(ci.primal_objective - ci.dual_objective)
    if optimality_norm == L_INF:
        primal_err = ci.l_inf_primal_residual
        primal_err_baseline = qp_cache.l_inf_norm_primal_right_hand_side
        dual_err = ci.l_inf_dual_residual
        dual_err_baseline = qp_cache.l_inf_norm_primal_linear_objective
    elif optimality_norm == L2:
        primal_err = ci.l2_primal_residual
        primal_err_baseline = qp_cache.l2_norm_primal_right_hand_side
        dual_err = ci.l2_dual_residual
        dual_err_baseline = qp_cache.l2_norm_primal_linear_objective
    else:
        raise ValueError("Unknown optimality_norm")
    return (dual_err < abs_tol + rel_tol * dual_err_baseline and
            primal_err < abs_tol + rel_tol * primal_err_baseline and
            gap < abs_tol + rel_tol * abs_obj)


def primal_infeasibility_criteria_met(eps_primal_infeasible, ii):
    """termination.jl:198-208"""
    if ii.dual_ray_objective <= 0.0:
        return False
    return ii.max_dual_ray_infeasibility / ii.dual_ray_objective <= eps_primal_infeasible


def dual_infeasibility_criteria_met(eps_dual_infeasible, ii):
    """termination.jl:213-227"""
    if ii.primal_ray_linear_objective >= 0.0:
        return False
    return (ii.max_primal_ray_infeasibility / (-ii.primal_ray_linear_objective)
            <= eps_dual_infeasible and
            ii.primal_ray_quadratic_norm / (-ii.primal_ray_linear_objective)
            <= eps_dual_infeasible)


def check_termination_criteria(criteria, qp_cache, iteration_stats):
    """termination.jl:233-273: a TerminationReason, or False to continue."""
    for ci in iteration_stats.convergence_information:
        if optimality_criteria_met(criteria.optimality_norm,
                                   criteria.eps_optimal_absolute,
                                   criteria.eps_optimal_relative, ci, qp_cache):
            return TerminationReason.TERMINATION_REASON_OPTIMAL
    for ii in iteration_stats.infeasibility_information:
        if primal_infeasibility_criteria_met(criteria.eps_primal_infeasible, ii):
            return TerminationReason.TERMINATION_REASON_PRIMAL_INFEASIBLE
        if dual_infeasibility_criteria_met(criteria.eps_dual_infeasible, ii):
            return TerminationReason.TERMINATION_REASON_DUAL_INFEASIBLE
    if iteration_stats.iteration_number >= criteria.iteration_limit:
        return TerminationReason.TERMINATION_REASON_ITERATION_LIMIT
    elif iteration_stats.cumulative_kkt_matrix_passes >= criteria.kkt_matrix_pass_limit:
        return TerminationReason.TERMINATION_REASON_KKT_MATRIX_PASS_LIMIT
    elif iteration_stats.cumulative_time_sec >= criteria.time_sec_limit:
        return TerminationReason.TERMINATION_REASON_TIME_LIMIT
    return False


def termination_reason_to_string(termination_reason):
    """termination.jl:275-277: strip the 'TERMINATION_REASON_' prefix."""
    return termination_reason.name[len("TERMINATION_REASON_"):]
