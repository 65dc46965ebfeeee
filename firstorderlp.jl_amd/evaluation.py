"""The evaluation / restart branch of ``optimize`` (pdhg.jl:892-1023) behind one
interface with two implementations:

* ``DeviceEvaluator`` -- everything reduced to scalars on the GPU through the
  C ABI (``pdhg_eval_point``, ``pdhg_trust_region_bound``,
  ``pdhg_distance_to_restart`` ...); no n-/m-length vector crosses PCIe until
  the final solution is fetched.  LP only.
* ``HostEvaluator`` -- numpy on host copies of the iterates, mat-vecs through
  ``ops`` (device SpMVs when the engine has them).  Used for QPs, for the
  row-partitioned engine and when the CPU oracle is injected in tests.

Points are ``_lib.POINT_CURRENT / POINT_AVERAGE / POINT_RESTART``.
"""
import math

import numpy as np

from . import _lib
from .iteration_stats_utils import evaluate_unscaled_iteration_stats
from .solve_log import (ConvergenceInformation, InfeasibilityInformation,
                        IterationStats)
from .trust_region_utils import (EUCLIDEAN_NORM, MAX_NORM,
                                 OptimalObjectiveBoundResult,
                                 bound_optimal_objective)

POINT_CURRENT, POINT_AVERAGE, POINT_RESTART = (_lib.POINT_CURRENT, _lib.POINT_AVERAGE,
                                               _lib.POINT_RESTART)


def _div(a, b):
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(np.float64(a) / np.float64(b))


class HostEvaluator:
    def __init__(self, engine, scaled_problem, qp_cache, ops, original_ops):
        self.engine = engine
        self.scaled_problem = scaled_problem
        self.problem = scaled_problem.scaled_qp
        self.qp_cache = qp_cache
        self.ops = ops
        self.original_ops = original_ops
        x0, y0 = engine.get_current()
        self.x_r, self.y_r = x0.copy(), y0.copy()   # create_last_restart_info, pdhg.jl:869

    def solution(self, point):
        if point == POINT_CURRENT:
            return self.engine.get_current()
        if point == POINT_AVERAGE:
            return self.engine.get_average()
        return self.x_r, self.y_r

    def iteration_stats(self, point, termination_criteria, record_iteration_stats, iteration,
                        cumulative_time, cumulative_kkt_passes, step_size, primal_weight,
                        candidate_type):
        x, y = self.solution(point)
        return evaluate_unscaled_iteration_stats(
            self.scaled_problem, self.qp_cache, termination_criteria, record_iteration_stats,
            x, y, iteration, cumulative_time, cumulative_kkt_passes,
            termination_criteria.eps_optimal_absolute, termination_criteria.eps_optimal_relative,
            step_size, primal_weight, candidate_type, self.original_ops)

    def point_sumsq(self, point):
        x, y = self.solution(point)
        return float(x @ x), float(y @ y)

    def distance_sq_to_restart(self, point):
        x, y = self.solution(point)
        dx, dy = x - self.x_r, y - self.y_r
        return float(dx @ dx), float(dy @ dy)

    def bound(self, point, primal_w, dual_w, radius, norm, approximate=False):
        x, y = self.solution(point)
        n, m = len(x), len(y)
        return bound_optimal_objective(self.problem, x, y, np.full(n, primal_w), np.full(m, dual_w),
                                       radius, norm, self.ops, solve_approximately=approximate)

    def restart(self, reset_to_average):
        """current .= avg (if requested), reset the average, remember the restart point."""
        if reset_to_average:
            self.engine.restart_to_average()
        self.engine.reset_average()
        x, y = self.engine.get_current()
        self.x_r, self.y_r = x.copy(), y.copy()

    def bounds(self, requests, primal_w, dual_w, norm, approximate=False):
        """[(point, radius), ...] -> [bound(...), ...] (the device evaluator batches these into one launch)."""
        return [self.bound(point, primal_w, dual_w, radius, norm, approximate) for point, radius in requests]



class DeviceEvaluator:
    def __init__(self, engine, scaled_problem, qp_cache):
        self.engine = engine
        self.qp_cache = qp_cache
        o = scaled_problem.original_qp
        self.objective_constant_original = o.objective_constant
        self.objective_constant_scaled = scaled_problem.scaled_qp.objective_constant
        self.scaled_problem = scaled_problem
        engine.set_original_problem(scaled_problem.constraint_rescaling,
                                    scaled_problem.variable_rescaling, o.objective_vector,
                                    o.right_hand_side, o.variable_lower_bound,
                                    o.variable_upper_bound)

    def solution(self, point):
        return self.engine.get_point(point)

    def iteration_stats(self, point, termination_criteria, record_iteration_stats, iteration,
                        cumulative_time, cumulative_kkt_passes, step_size, primal_weight,
                        candidate_type):
        """compute_iteration_stats (iteration_stats_utils.jl:356-407) assembled
        from the device's raw sums/maxes (include/pdhg_hip.h, pdhg_eval_point)."""
        r = self.engine.eval_point(point)
        (S0, S1, S2, S3, M0, M1, M2, M3,
         T0, T1, T2, T3, T4, T5, N0, N1, N2, N3, N4, N5) = [float(v) for v in r[:20]]
        xqx, max_qx = float(r[20]), float(r[21])        # x'Qx and |Qx|_inf (0 for an LP)
        qp = self.qp_cache
        eps_ratio = _div(termination_criteria.eps_optimal_absolute,
                         termination_criteria.eps_optimal_relative)
        ci = ConvergenceInformation()
        # primal_obj: c'x + 0.5 x'Qx + constant                    iteration_stats_utils.jl:66-75
        ci.primal_objective = self.objective_constant_original + T3 + 0.5 * xqx
        ci.l_inf_primal_residual = max(M0, N2)
        ci.l2_primal_residual = math.sqrt(S0 + T4)
        ci.relative_l_inf_primal_residual = _div(ci.l_inf_primal_residual,
                                                 eps_ratio + qp.l_inf_norm_primal_right_hand_side)
        ci.relative_l2_primal_residual = _div(ci.l2_primal_residual,
                                              eps_ratio + qp.l2_norm_primal_right_hand_side)
        ci.l_inf_primal_variable = N1
        ci.l2_primal_variable = math.sqrt(T2)
        # base dual objective b'y + constant - 0.5 x'Qx, plus bound*rc   iteration_stats_utils.jl:180-196
        ci.dual_objective = (S2 + self.objective_constant_original - 0.5 * xqx) + T1
        ci.l_inf_dual_residual = max(M3, N0)
        ci.l2_dual_residual = math.sqrt(S3 + T0)
        ci.relative_l_inf_dual_residual = _div(ci.l_inf_dual_residual,
                                               eps_ratio + qp.l_inf_norm_primal_linear_objective)
        ci.relative_l2_dual_residual = _div(ci.l2_dual_residual,
                                            eps_ratio + qp.l2_norm_primal_linear_objective)
        ci.l_inf_dual_variable = M2
        ci.l2_dual_variable = math.sqrt(S1)
        ci.corrected_dual_objective = ci.dual_objective if ci.l_inf_dual_residual == 0.0 else -math.inf
        gap = abs(ci.primal_objective - ci.dual_objective)
        abs_obj = abs(ci.primal_objective) + abs(ci.dual_objective)
        ci.relative_optimality_gap = _div(gap, eps_ratio + abs_obj)
        ci.candidate_type = candidate_type

        ii = InfeasibilityInformation()
        s = N1 if N1 != 0.0 else 1.0                    # primal ray scaled to unit inf-norm
        ii.max_primal_ray_infeasibility = max(M1, N5) / s
        ii.primal_ray_linear_objective = T3 / s
        ii.primal_ray_quadratic_norm = max_qx / s       # |Q ray|_inf     iteration_stats_utils.jl:316-317
        scaling_factor = max(M2, N4)
        if scaling_factor != 0.0:
            ii.max_dual_ray_infeasibility = max(M3, N3) / scaling_factor
            ii.dual_ray_objective = (S2 + T5) / scaling_factor
        ii.candidate_type = candidate_type

        stats = IterationStats()
        stats.iteration_number = int(iteration - 1)
        stats.cumulative_kkt_matrix_passes = cumulative_kkt_passes
        stats.cumulative_time_sec = cumulative_time
        stats.convergence_information = [ci]
        stats.infeasibility_information = [ii]
        stats.step_size = step_size
        stats.primal_weight = primal_weight
        stats.method_specific_stats = {}
        return stats

    def point_sumsq(self, point):
        return self.engine.point_sumsq(point)

    def distance_sq_to_restart(self, point):
        return self.engine.distance_to_restart(point)

    def bound(self, point, primal_w, dual_w, radius, norm, approximate=False):
        """bound_optimal_objective (trust_region_utils.jl:271-360)."""
        const = self.objective_constant_scaled
        if norm == EUCLIDEAN_NORM:
            o = self.engine.trust_region_bound(point, primal_w, dual_w, radius, 0, approximate)
            lag = float(o[0]) + const
            return OptimalObjectiveBoundResult(lag, lag + float(o[1]), lag - float(o[2]), None, None)
        op = self.engine.trust_region_bound(point, primal_w, dual_w, radius, 1, approximate)
        od = self.engine.trust_region_bound(point, primal_w, dual_w, radius, 2, approximate)
        lag = float(op[0]) + const
        return OptimalObjectiveBoundResult(lag, lag + float(op[1]), lag - float(od[2]), None, None)

    def bounds(self, requests, primal_w, dual_w, norm, approximate=False):
        """Several bound_optimal_objective problems at once: requests = [(point, radius), ...] -> one
        OptimalObjectiveBoundResult each, the very numbers ``bound`` returns one by one (pdhg_trust_region_bounds: on
        medium single handles the searches share one persistent launch)."""
        const = self.objective_constant_scaled
        if norm == EUCLIDEAN_NORM:
            out = []
            for i in range(0, len(requests), 3):
                chunk = requests[i:i + 3]
                rows = self.engine.trust_region_bounds([p for p, _ in chunk], primal_w, dual_w, [r for _, r in chunk],
                                                       [0] * len(chunk), approximate)
                for o in rows:
                    lag = float(o[0]) + const
                    out.append(OptimalObjectiveBoundResult(lag, lag + float(o[1]), lag - float(o[2]), None, None))
            return out
        out = []
        for point, radius in requests:           # MAX_NORM: the primal and the dual half of one point share a launch
            op, od = self.engine.trust_region_bounds([point, point], primal_w, dual_w, [radius, radius], [1, 2], approximate)
            lag = float(op[0]) + const
            out.append(OptimalObjectiveBoundResult(lag, lag + float(op[1]), lag - float(od[2]), None, None))
        return out

    def restart(self, reset_to_average):
        if reset_to_average:
            self.engine.restart_to_average()
        self.engine.reset_average()
        self.engine.save_restart_point()
