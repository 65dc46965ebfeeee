"""Bound-constrained trust region / localized duality gap: host logic
mirroring src/trust_region_utils.jl (evaluation cadence only; sort/median
heavy, not bandwidth critical)."""
import enum
import math
from dataclasses import dataclass

import numpy as np


def weighted_norm(vec, weights):
    """saddle_point.jl:120-129: sqrt(sum w_i v_i^2)."""
    return math.sqrt(float(np.sum(weights * vec * vec)))


@dataclass
class BoundConstrainedTrustRegionResult:
    """trust_region_utils.jl:21-26"""
    solution: np.ndarray
    value: float


def _julia_median(v):
    """Statistics.median: mean of the two middle order statistics."""
    return float(np.median(v))


def solve_bound_constrained_trust_region(center_point, objective_vector,
                                         variable_lower_bounds,
                                         variable_upper_bounds, norm_weights,
                                         target_radius, solve_approximately):
    """trust_region_utils.jl:68-192 (median-of-breakpoints search)."""
    if solve_approximately:
        return approximately_solve_bound_constrained_trust_region(
            center_point, objective_vector, variable_lower_bounds,
            variable_upper_bounds, norm_weights, target_radius)
    assert 0.0 <= target_radius < math.inf
    if target_radius == 0.0 or float(np.sum(objective_vector ** 2)) == 0.0:
        return BoundConstrainedTrustRegionResult(center_point.copy(), 0.0)

    with np.errstate(divide="ignore", invalid="ignore"):
        fixed = ((center_point >= variable_upper_bounds) & (objective_vector <= 0)) | \
                ((center_point <= variable_lower_bounds) & (objective_vector >= 0))
        direction = np.where(fixed, 0.0, -objective_vector / norm_weights)
        threshold = np.zeros_like(center_point)
        pos = direction > 0
        neg = direction < 0
        threshold[pos] = (variable_upper_bounds[pos] - center_point[pos]) / direction[pos]
        threshold[neg] = (variable_lower_bounds[neg] - center_point[neg]) / direction[neg]

    low_radius_sq = 0.0
    high_radius_sq = 0.0
    infinite = np.isinf(threshold)
    high_radius_sq += weighted_norm(direction[infinite], norm_weights[infinite]) ** 2
    indices = np.nonzero(np.isfinite(threshold))[0]

    while len(indices) > 0:
        th = threshold[indices]
        test_threshold = _julia_median(th)
        test_point = np.clip(center_point[indices] + test_threshold * direction[indices],
                             variable_lower_bounds[indices], variable_upper_bounds[indices])
        test_radius = weighted_norm(test_point - center_point[indices], norm_weights[indices])
        if low_radius_sq + test_radius ** 2 + test_threshold ** 2 * high_radius_sq >= target_radius ** 2:
            discard = indices[th >= test_threshold]
            high_radius_sq += weighted_norm(direction[discard], norm_weights[discard]) ** 2
            indices = indices[th < test_threshold]
        else:
            discard = indices[th <= test_threshold]
            discard_point = np.clip(center_point[discard] + test_threshold * direction[discard],
                                    variable_lower_bounds[discard], variable_upper_bounds[discard])
            low_radius_sq += weighted_norm(discard_point - center_point[discard],
                                           norm_weights[discard]) ** 2
            indices = indices[th > test_threshold]

    if high_radius_sq <= 0.0:
        target_threshold = float(np.max(threshold))
    else:
        target_threshold = math.sqrt((target_radius ** 2 - low_radius_sq) / high_radius_sq)
    candidate_point = np.clip(center_point + target_threshold * direction,
                              variable_lower_bounds, variable_upper_bounds)
    return BoundConstrainedTrustRegionResult(
        candidate_point, float(objective_vector @ (candidate_point - center_point)))


def approximately_solve_bound_constrained_trust_region(center_point, objective_vector,
                                                       variable_lower_bounds,
                                                       variable_upper_bounds,
                                                       norm_weights, target_radius):
    """trust_region_utils.jl:194-224"""
    fixed = ((center_point >= variable_upper_bounds) & (objective_vector <= 0)) | \
            ((center_point <= variable_lower_bounds) & (objective_vector >= 0))
    direction = np.where(fixed, 0.0, -objective_vector / norm_weights)
    direction_norm = weighted_norm(direction, norm_weights)
    if direction_norm > 0.0:
        direction = direction * (target_radius / direction_norm)
    return BoundConstrainedTrustRegionResult(center_point + direction,
                                             float(objective_vector @ direction))


@dataclass
class OptimalObjectiveBoundResult:
    """trust_region_utils.jl:226-234"""
    lagrangian_value: float
    lower_bound_value: float
    upper_bound_value: float
    primal_solution: np.ndarray
    dual_solution: np.ndarray


def get_gap(result):
    """trust_region_utils.jl:236-238"""
    return result.upper_bound_value - result.lower_bound_value


class LocalizedDualityGapNorm(enum.Enum):
    """trust_region_utils.jl:245"""
    MAX_NORM = 0
    EUCLIDEAN_NORM = 1


MAX_NORM = LocalizedDualityGapNorm.MAX_NORM
EUCLIDEAN_NORM = LocalizedDualityGapNorm.EUCLIDEAN_NORM


def bound_optimal_objective(problem, primal_solution, dual_solution,
                            primal_norm_weights, dual_norm_weights,
                            distance_to_optimality, norm, ops,
                            solve_approximately=False):
    """trust_region_utils.jl:271-360.  ``ops`` supplies A*x, A'*y, Q*x on the
    (scaled) problem -- device SpMVs in the product path."""
    n = len(primal_solution)
    aty = ops.ATy(dual_solution)
    qx = ops.Qx(primal_solution)
    # compute_primal_gradient                         saddle_point.jl:1081-1100
    primal_gradient = qx + problem.objective_vector - aty
    # compute_lagrangian_value                        saddle_point.jl:1109-1120
    lagrangian_value = (0.5 * float(primal_solution @ qx) +
                        float(primal_solution @ problem.objective_vector) -
                        float(primal_solution @ aty) +
                        float(dual_solution @ problem.right_hand_side) +
                        problem.objective_constant)
    m = len(dual_solution)
    dual_variable_lower_bounds = np.full(m, -np.inf)
    dual_variable_upper_bounds = np.full(m, np.inf)
    dual_variable_lower_bounds[problem.num_equalities:] = 0.0
    # compute_dual_gradient                            saddle_point.jl:1102-1107
    dual_gradient = problem.right_hand_side - ops.Ax(primal_solution)

    if norm == MAX_NORM:
        primal_result = solve_bound_constrained_trust_region(
            primal_solution, primal_gradient, problem.variable_lower_bound,
            problem.variable_upper_bound, primal_norm_weights,
            distance_to_optimality, solve_approximately)
        dual_result = solve_bound_constrained_trust_region(
            dual_solution, -dual_gradient, dual_variable_lower_bounds,
            dual_variable_upper_bounds, dual_norm_weights,
            distance_to_optimality, solve_approximately)
        return OptimalObjectiveBoundResult(
            lagrangian_value, lagrangian_value + primal_result.value,
            lagrangian_value - dual_result.value, primal_result.solution,
            dual_result.solution)
    elif norm == EUCLIDEAN_NORM:
        z = np.concatenate([primal_solution, dual_solution])
        z_gradient = np.concatenate([primal_gradient, -dual_gradient])
        z_lower_bound = np.concatenate([problem.variable_lower_bound, dual_variable_lower_bounds])
        z_upper_bound = np.concatenate([problem.variable_upper_bound, dual_variable_upper_bounds])
        norm_weights = np.concatenate([primal_norm_weights, dual_norm_weights])
        result = solve_bound_constrained_trust_region(
            z, z_gradient, z_lower_bound, z_upper_bound, norm_weights,
            distance_to_optimality, solve_approximately)
        primal_tr_solution = result.solution[:n]
        dual_tr_solution = result.solution[n:]
        return OptimalObjectiveBoundResult(
            lagrangian_value,
            lagrangian_value + float((primal_tr_solution - primal_solution) @ primal_gradient),
            lagrangian_value + float((dual_tr_solution - dual_solution) @ dual_gradient),
            primal_tr_solution, dual_tr_solution)
    raise ValueError(f"unknown norm = {norm}, value unknown")
