"""Restart scheme, primal-weight update and output un-scaling: host control
logic mirroring src/saddle_point.jl (the parts that are NOT on the per-iteration
hot path; the hot-path parts -- projections, weighted average, gradients --
live in csrc/pdhg_hip.hip)."""
import enum
import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from .solve_log import IterationStats, RestartChoice, TerminationReason
from .termination import termination_reason_to_string
from .trust_region_utils import (EUCLIDEAN_NORM, MAX_NORM,
                                 OptimalObjectiveBoundResult,
                                 bound_optimal_objective, get_gap,
                                 weighted_norm)

EPS = float(np.finfo(np.float64).eps)


def _div(a, b):
    """Julia float division (x/0 -> Inf/NaN, no exception)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(np.float64(a) / np.float64(b))


@dataclass
class SaddlePointOutput:
    """saddle_point.jl:22-53"""
    primal_solution: np.ndarray
    dual_solution: np.ndarray
    termination_reason: TerminationReason
    termination_string: str
    iteration_count: int
    iteration_stats: List[IterationStats]


def unscaled_saddle_point_output(scaled_problem, primal_solution, dual_solution,
                                 termination_reason, iterations_completed,
                                 iteration_stats):
    """saddle_point.jl:55-77"""
    return SaddlePointOutput(
        primal_solution / scaled_problem.variable_rescaling,
        dual_solution / scaled_problem.constraint_rescaling,
        termination_reason, termination_reason_to_string(termination_reason),
        int(iterations_completed), iteration_stats)


class RestartScheme(enum.Enum):
    """saddle_point.jl:322"""
    NO_RESTARTS = 0
    FIXED_FREQUENCY = 1
    ADAPTIVE_NORMALIZED = 2
    ADAPTIVE_LOCALIZED = 3
    ADAPTIVE_DISTANCE = 4


class RestartToCurrentMetric(enum.Enum):
    """saddle_point.jl:337"""
    NO_RESTART_TO_CURRENT = 0
    GAP_OVER_DISTANCE = 1
    GAP_OVER_DISTANCE_SQUARED = 2


for _e in (RestartScheme, RestartToCurrentMetric):
    for _m in _e:
        globals()[_m.name] = _m


@dataclass
class RestartParameters:
    """saddle_point.jl:339-400"""
    restart_scheme: RestartScheme
    restart_to_current_metric: RestartToCurrentMetric
    restart_frequency_if_fixed: int
    artificial_restart_threshold: float
    sufficient_reduction_for_restart: float
    necessary_reduction_for_restart: float
    primal_weight_update_smoothing: float
    use_approximate_localized_duality_gap: bool


def construct_restart_parameters(restart_scheme, restart_to_current_metric,
                                 restart_frequency_if_fixed,
                                 artificial_restart_threshold,
                                 sufficient_reduction_for_restart,
                                 necessary_reduction_for_restart,
                                 primal_weight_update_smoothing,
                                 use_approximate_localized_duality_gap):
    """saddle_point.jl:402-430"""
    assert restart_frequency_if_fixed > 1
    assert 0.0 < artificial_restart_threshold <= 1.0
    assert 0.0 < sufficient_reduction_for_restart <= necessary_reduction_for_restart <= 1.0
    assert 0.0 <= primal_weight_update_smoothing <= 1.0
    return RestartParameters(restart_scheme, restart_to_current_metric,
                             restart_frequency_if_fixed,
                             artificial_restart_threshold,
                             sufficient_reduction_for_restart,
                             necessary_reduction_for_restart,
                             primal_weight_update_smoothing,
                             use_approximate_localized_duality_gap)


@dataclass
class RestartInfo:
    """saddle_point.jl:158-198.  The restart point's vectors
    (primal_solution / dual_solution) are held by the evaluator -- on the
    device in the product path."""
    last_restart_localized_duality_gap: Optional[OptimalObjectiveBoundResult]
    last_restart_length: int
    primal_distance_moved_last_restart_period: float
    dual_distance_moved_last_restart_period: float
    gap_reduction_ratio_last_trial: float


def create_last_restart_info():
    """saddle_point.jl:200-213 (the point itself is the evaluator's restart point)."""
    return RestartInfo(None, 1, 0.0, 0.0, 1.0)


def compute_localized_duality_gaps(ev, primal_weight_norm, dual_weight_norm,
                                   use_approximate_localized_duality_gap, extra_request=None):
    """saddle_point.jl:432-496.  ``ev`` is an evaluator (evaluation.py); the norm
    weights are uniform per block in PDHG (define_norms, pdhg.jl:265-277), so
    weighted_norm(v, w)^2 == w * sum(v^2).
    The two trust-region problems (and ``extra_request`` = (point, radius), the bound at the last restart point that
    should_do_adaptive_restart_normalized_duality_gap is about to ask for) go to the evaluator as ONE request: on the
    device their searches share one persistent launch (pdhg_trust_region_bounds); each result is what the call on its
    own returns."""
    from .evaluation import POINT_AVERAGE, POINT_CURRENT
    dx2, dy2 = ev.distance_sq_to_restart(POINT_AVERAGE)
    distance_traveled_by_average = math.sqrt(primal_weight_norm * dx2 + dual_weight_norm * dy2)
    cx2, cy2 = ev.distance_sq_to_restart(POINT_CURRENT)
    distance_traveled_by_current = math.sqrt(primal_weight_norm * cx2 + dual_weight_norm * cy2)
    requests = [(POINT_AVERAGE, distance_traveled_by_average), (POINT_CURRENT, distance_traveled_by_current)]
    if extra_request is not None:
        requests.append(extra_request)
    got = ev.bounds(requests, primal_weight_norm, dual_weight_norm, EUCLIDEAN_NORM,
                    use_approximate_localized_duality_gap)
    return dict(gap_at_average=got[0],
                distance_traveled_by_average=distance_traveled_by_average,
                gap_at_current=got[1],
                distance_traveled_by_current=distance_traveled_by_current,
                average_distance_sq=(dx2, dy2),
                extra=got[2] if extra_request is not None else None)


def should_reset_to_average(current, distance_traveled_by_current, average,
                            distance_traveled_by_average,
                            restart_to_current_metric):
    """saddle_point.jl:530-549"""
    current_normalized_gap = _div(get_gap(current), distance_traveled_by_current)
    average_normalized_gap = _div(get_gap(average), distance_traveled_by_average)
    if restart_to_current_metric == RestartToCurrentMetric.GAP_OVER_DISTANCE_SQUARED:
        return _div(current_normalized_gap, distance_traveled_by_current) >= \
            _div(average_normalized_gap, distance_traveled_by_average)
    elif restart_to_current_metric == RestartToCurrentMetric.GAP_OVER_DISTANCE:
        return current_normalized_gap >= average_normalized_gap
    return True


def distance_traveled_in_last_restart_period(last_restart_info, primal_weight):
    """saddle_point.jl:561-565"""
    lri = last_restart_info
    return math.sqrt(lri.primal_distance_moved_last_restart_period ** 2 * primal_weight +
                     lri.dual_distance_moved_last_restart_period ** 2 / primal_weight)


def should_do_adaptive_restart_normalized_duality_gap(
        ev, primal_weight_norm, dual_weight_norm, candidate_localized_gap,
        candidate_distance_traveled, restart_params, last_restart_info,
        use_approximate_localized_duality_gap, primal_weight, last_restart_bound=None):
    """saddle_point.jl:551-596.  ``last_restart_bound``: the bound at the last restart point when the caller already
    has it (run_restart_scheme asks for it together with the two candidate bounds)."""
    from .evaluation import POINT_RESTART
    lri = last_restart_info
    distance_traveled_last_restart = distance_traveled_in_last_restart_period(lri, primal_weight)
    last_restart = last_restart_bound if last_restart_bound is not None else \
        ev.bound(POINT_RESTART, primal_weight_norm, dual_weight_norm,
                 distance_traveled_last_restart, EUCLIDEAN_NORM,
                 use_approximate_localized_duality_gap)
    do_restart = False
    normalized_candidate_gap = _div(get_gap(candidate_localized_gap), candidate_distance_traveled)
    normalized_last_restart_gap = _div(get_gap(last_restart), distance_traveled_last_restart)
    gap_reduction_ratio = _div(normalized_candidate_gap, normalized_last_restart_gap)
    if gap_reduction_ratio < restart_params.necessary_reduction_for_restart:
        if gap_reduction_ratio < restart_params.sufficient_reduction_for_restart:
            do_restart = True
        elif gap_reduction_ratio > lri.gap_reduction_ratio_last_trial:
            do_restart = True
    lri.gap_reduction_ratio_last_trial = gap_reduction_ratio
    return do_restart


def should_do_localized_adaptive_restart(candidate_localized_gap,
                                         candidate_restart_length,
                                         restart_params, last_restart_info):
    """saddle_point.jl:600-624"""
    lri = last_restart_info
    if candidate_localized_gap is None or lri.last_restart_localized_duality_gap is None:
        return True
    new_potential = _div(get_gap(candidate_localized_gap), candidate_restart_length)
    old_potential = _div(get_gap(lri.last_restart_localized_duality_gap), lri.last_restart_length)
    return _div(new_potential, old_potential) < restart_params.necessary_reduction_for_restart


def should_do_distance_based_adaptive_restart(candidate_localized_gap,
                                              candidate_distance_traveled,
                                              candidate_restart_length,
                                              restart_params, last_restart_info,
                                              primal_weight):
    """saddle_point.jl:627-653"""
    lri = last_restart_info
    distance_traveled_last_restart = math.sqrt(
        lri.primal_distance_moved_last_restart_period ** 2 * primal_weight +
        lri.dual_distance_moved_last_restart_period ** 2 / primal_weight)
    new_potential = _div(candidate_distance_traveled, candidate_restart_length)
    old_potential = _div(distance_traveled_last_restart, lri.last_restart_length)
    return _div(new_potential, old_potential) < restart_params.necessary_reduction_for_restart


def run_restart_scheme(ev, last_restart_info, iterations_completed,
                       primal_weight_norm, dual_weight_norm, primal_weight,
                       verbosity, restart_params):
    """saddle_point.jl:688-846.  ``ev`` (evaluation.py) owns the vector work:
    solution_weighted_avg, the current iterate and the last restart point live
    on the device; a restart to the average is performed there
    (pdhg_restart_to_average, which also recomputes A'y as pdhg.jl:1018-1022
    does).  ``primal_weight_norm`` / ``dual_weight_norm`` are the uniform
    entries of define_norms' vectors.  Returns a RestartChoice."""
    from .evaluation import POINT_AVERAGE
    engine = ev.engine
    count_x, count_y, _, _ = engine.average_info()
    if not (count_x > 0 and count_y > 0):
        return RestartChoice.RESTART_CHOICE_NO_RESTART

    restart_length = count_x
    artificial_restart = False
    do_restart = False
    if restart_length >= restart_params.artificial_restart_threshold * iterations_completed:
        do_restart = True
        artificial_restart = True

    average_distance_sq = None
    if restart_params.restart_scheme == RestartScheme.NO_RESTARTS:
        reset_to_average = False
        candidate_localized_gap = None
        candidate_distance_traveled = None
    else:
        # (the bound at the last restart point, which the adaptive-normalized test below needs, rides in the same request)
        extra = None
        if not do_restart and restart_params.restart_scheme == RestartScheme.ADAPTIVE_NORMALIZED:
            from .evaluation import POINT_RESTART
            extra = (POINT_RESTART, distance_traveled_in_last_restart_period(last_restart_info, primal_weight))
        gaps = compute_localized_duality_gaps(
            ev, primal_weight_norm, dual_weight_norm,
            restart_params.use_approximate_localized_duality_gap, extra)
        average_distance_sq = gaps["average_distance_sq"]
        reset_to_average = should_reset_to_average(
            gaps["gap_at_current"], gaps["distance_traveled_by_current"],
            gaps["gap_at_average"], gaps["distance_traveled_by_average"],
            restart_params.restart_to_current_metric)
        if reset_to_average:
            candidate_localized_gap = gaps["gap_at_average"]
            candidate_distance_traveled = gaps["distance_traveled_by_average"]
        else:
            candidate_localized_gap = gaps["gap_at_current"]
            candidate_distance_traveled = gaps["distance_traveled_by_current"]

    if not do_restart:
        scheme = restart_params.restart_scheme
        if scheme == RestartScheme.ADAPTIVE_NORMALIZED:
            do_restart = should_do_adaptive_restart_normalized_duality_gap(
                ev, primal_weight_norm, dual_weight_norm,
                candidate_localized_gap, candidate_distance_traveled,
                restart_params, last_restart_info,
                restart_params.use_approximate_localized_duality_gap, primal_weight, gaps["extra"])
        elif scheme in (RestartScheme.ADAPTIVE_LOCALIZED, RestartScheme.ADAPTIVE_DISTANCE) and \
                last_restart_info.last_restart_localized_duality_gap is None:
            do_restart = True
        elif scheme == RestartScheme.ADAPTIVE_LOCALIZED:
            do_restart = should_do_localized_adaptive_restart(
                candidate_localized_gap, restart_length, restart_params, last_restart_info)
        elif scheme == RestartScheme.ADAPTIVE_DISTANCE:
            do_restart = should_do_distance_based_adaptive_restart(
                candidate_localized_gap, candidate_distance_traveled,
                restart_length, restart_params, last_restart_info, primal_weight)
        elif scheme == RestartScheme.FIXED_FREQUENCY and \
                restart_params.restart_frequency_if_fixed <= restart_length:
            do_restart = True

    if not do_restart:
        return RestartChoice.RESTART_CHOICE_NO_RESTART

    if verbosity >= 4:
        print("  Restarted to average" if reset_to_average else "  Restarted to current",
              " after ", str(restart_length).ljust(4), " iterations",
              "*" if artificial_restart else "", sep="")
    # update_last_restart_info (saddle_point.jl:893-927): distances use the
    # average against the OLD restart point, so take them before restarting.
    if average_distance_sq is None:
        average_distance_sq = ev.distance_sq_to_restart(POINT_AVERAGE)
    lri = last_restart_info
    lri.primal_distance_moved_last_restart_period = \
        math.sqrt(primal_weight_norm * average_distance_sq[0]) / math.sqrt(primal_weight)
    lri.dual_distance_moved_last_restart_period = \
        math.sqrt(dual_weight_norm * average_distance_sq[1]) * math.sqrt(primal_weight)
    lri.last_restart_length = restart_length
    lri.last_restart_localized_duality_gap = candidate_localized_gap
    # current .= avg (if chosen) ; reset_solution_weighted_average ; restart point .= current
    ev.restart(reset_to_average)
    if reset_to_average:
        return RestartChoice.RESTART_CHOICE_RESTART_TO_AVERAGE
    return RestartChoice.RESTART_CHOICE_WEIGHTED_AVERAGE_RESET


def compute_new_primal_weight(last_restart_info, primal_weight,
                              primal_weight_update_smoothing, verbosity):
    """saddle_point.jl:862-891"""
    primal_distance = last_restart_info.primal_distance_moved_last_restart_period
    dual_distance = last_restart_info.dual_distance_moved_last_restart_period
    if primal_distance > EPS and dual_distance > EPS:
        new_primal_weight_estimate = dual_distance / primal_distance
        log_primal_weight = (
            primal_weight_update_smoothing * math.log(new_primal_weight_estimate) +
            (1 - primal_weight_update_smoothing) * math.log(primal_weight))
        primal_weight = math.exp(log_primal_weight)
        if verbosity >= 4:
            print("  New computed primal weight is %.2e" % primal_weight)
        return primal_weight
    return primal_weight


def update_objective_bound_estimates(method_specific_stats, ev, point,
                                     primal_weight_norm, dual_weight_norm):
    """saddle_point.jl:1015-1047 (uniform norm weights)."""
    sx2, sy2 = ev.point_sumsq(point)
    estimated_primal_distance_to_optimality = max(1e-8, math.sqrt(primal_weight_norm * sx2))
    estimated_dual_distance_to_optimality = max(1e-8, math.sqrt(dual_weight_norm * sy2))
    # (through `bounds`: the primal and the dual half of MAX_NORM are two trust-region problems, one launch on the device)
    gap = ev.bounds([(point, 1.0)],
                    _div(primal_weight_norm, estimated_primal_distance_to_optimality ** 2),
                    _div(dual_weight_norm, estimated_dual_distance_to_optimality ** 2),
                    MAX_NORM, False)[0]
    method_specific_stats["lagrangian_value"] = gap.lagrangian_value
    method_specific_stats["estimated_lower_bound"] = gap.lower_bound_value
    method_specific_stats["estimated_upper_bound"] = gap.upper_bound_value


def select_initial_primal_weight(problem, primal_norm_params, dual_norm_params,
                                 primal_importance, verbosity):
    """saddle_point.jl:1049-1075"""
    rhs_vec_norm = weighted_norm(problem.right_hand_side, dual_norm_params)
    obj_vec_norm = weighted_norm(problem.objective_vector, primal_norm_params)
    if obj_vec_norm > 0.0 and rhs_vec_norm > 0.0:
        primal_weight = primal_importance * (obj_vec_norm / rhs_vec_norm)
    else:
        primal_weight = primal_importance
    if verbosity >= 6:
        print(f"Initial primal weight = {primal_weight}")
    return primal_weight


def compute_lagrangian_value(problem, primal_solution, dual_solution, ops=None):
    """saddle_point.jl:1109-1120"""
    if ops is None:
        from .iteration_stats_utils import HostOps
        ops = HostOps(problem)
    return (0.5 * float(primal_solution @ ops.Qx(primal_solution)) +
            float(primal_solution @ problem.objective_vector) -
            float(primal_solution @ ops.ATy(dual_solution)) +
            float(dual_solution @ problem.right_hand_side) +
            problem.objective_constant)
