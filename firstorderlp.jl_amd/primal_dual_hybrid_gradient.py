"""Host driver of the PDHG inner loop: the reference's
src/primal_dual_hybrid_gradient.jl with the n-/m-length vector work moved
behind the C ABI (``engine.HipPdhgEngine``).

What stays on the host (as the reference's north star prescribes): the scalar
step-size rules of ``take_step`` (pdhg.jl:555-767), counters, and -- in
``optimize`` -- restarts, primal-weight updates and termination.
"""
import math
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np


# ---- step-size policy parameter structs (pdhg.jl:19-68) ----------------------

@dataclass
class MalitskyPockStepsizeParameters:
    """pdhg.jl:19-42"""
    downscaling_factor: float
    breaking_factor: float
    interpolation_coefficient: float


@dataclass
class AdaptiveStepsizeParams:
    """pdhg.jl:44-63"""
    reduction_exponent: float
    growth_exponent: float


@dataclass
class ConstantStepsizeParams:
    """pdhg.jl:65-68"""
    pass


@dataclass
class PdhgSolverState:
    """pdhg.jl:205-258.  The vector fields (current_primal_solution,
    current_dual_solution, delta_*, current_dual_product,
    solution_weighted_avg) live on the device inside ``engine``."""
    engine: object
    step_size: float = 0.0
    primal_weight: float = 1.0
    numerical_error: bool = False
    cumulative_kkt_passes: float = 0.0
    total_number_iterations: int = 0
    required_ratio: Optional[float] = None
    ratio_step_sizes: Optional[float] = None

    # convenience accessors (device -> host copies)
    @property
    def current_primal_solution(self):
        return self.engine.get_current()[0]

    @property
    def current_dual_solution(self):
        return self.engine.get_current()[1]

    @property
    def current_dual_product(self):
        return self.engine.get_dual_product()


def interaction_and_movement(raw, primal_weight):
    """compute_interaction_and_movement (pdhg.jl:527-549) from the device's raw
    sums: raw = [dx.(A'y'-A'y), sum dx^2, sum dy^2, sum dAty^2, 0.5 dx'Qdx].
    ``norm(v)^2`` is restated literally as sqrt(sum)^2."""
    interaction = abs(raw[0]) + abs(raw[4])
    nx = math.sqrt(raw[1])
    ny = math.sqrt(raw[2])
    movement = 0.5 * primal_weight * (nx * nx) + (0.5 / primal_weight) * (ny * ny)
    return interaction, movement


def julia_min(a, b):
    """Julia's ``min`` on Float64: a NaN operand gives NaN (Python's ``min`` returns
    whichever operand the comparison happens to favour).  pdhg.jl:729 uses Julia's: once a
    NaN reaches the step-size rule the reference's step size IS NaN from then on (and, since
    `step_size <= limit` is false for a NaN on either side, its retry loop never accepts
    again -- the reference relies on NaN-free data); the twins here keep the same scalars."""
    if a != a or b != b:
        return math.nan
    return a if a < b else b


def take_step_adaptive(step_params, solver_state):
    """take_step(::AdaptiveStepsizeParams, ...)  pdhg.jl:653-731.

    With the HIP engine the same statements run inside the library
    (pdhg_take_step_adaptive: one call per iteration instead of two with
    Python in between; bitwise the same scalars).  PDHG_PY_TAKE_STEP=1 keeps
    the loop below, which is also what every other engine uses."""
    eng = solver_state.engine
    if hasattr(eng, "take_step_adaptive") and os.environ.get("PDHG_PY_TAKE_STEP", "0") != "1":
        (solver_state.step_size, solver_state.total_number_iterations,
         solver_state.cumulative_kkt_passes, err) = eng.take_step_adaptive(
            step_params.reduction_exponent, step_params.growth_exponent, solver_state.step_size,
            solver_state.primal_weight, solver_state.total_number_iterations,
            solver_state.cumulative_kkt_passes)
        if err:
            solver_state.numerical_error = True
        return
    step_size = solver_state.step_size
    done = False
    while not done:
        solver_state.total_number_iterations += 1
        raw = eng.trial_step(step_size, solver_state.primal_weight, 1.0)
        interaction, movement = interaction_and_movement(
            raw, solver_state.primal_weight)
        solver_state.cumulative_kkt_passes += 1
        if movement == 0.0:
            # The algorithm will terminate at the beginning of the next iteration
            solver_state.numerical_error = True
            break
        if interaction > 0:
            step_size_limit = movement / interaction
        else:
            step_size_limit = math.inf
        if step_size <= step_size_limit:
            # update_solution_in_solver_state: weight = solver_state.step_size,
            # the value on entry to take_step (pdhg.jl:512)
            eng.accept(solver_state.step_size)
            done = True
        k1 = float(solver_state.total_number_iterations + 1)
        first_term = (1 - k1 ** (-step_params.reduction_exponent)) * step_size_limit
        second_term = (1 + k1 ** (-step_params.growth_exponent)) * step_size
        step_size = julia_min(first_term, second_term)
    solver_state.step_size = step_size


def take_step_constant(step_params, solver_state):
    """take_step(::ConstantStepsizeParams, ...)  pdhg.jl:737-767."""
    eng = solver_state.engine
    eng.trial_step(solver_state.step_size, solver_state.primal_weight, 1.0)
    solver_state.cumulative_kkt_passes += 1
    eng.accept(solver_state.step_size)


def take_step_malitsky_pock(step_params, solver_state, is_lp=True):
    """take_step(::MalitskyPockStepsizeParameters, ...)  pdhg.jl:555-647."""
    if not is_lp:
        raise ValueError("Malitsky and Pock linesearch is only supported for "
                         "linear programming problems.")
    eng = solver_state.engine
    step_size = solver_state.step_size
    ratio_step_sizes = solver_state.ratio_step_sizes
    done = False
    it = 0
    eng.trial_primal(step_size, solver_state.primal_weight)
    solver_state.cumulative_kkt_passes += 0.5
    step_size = step_size + step_params.interpolation_coefficient * \
        (math.sqrt(1 + ratio_step_sizes) - 1) * step_size
    max_iter = 60
    while not done and it < max_iter:
        it += 1
        solver_state.total_number_iterations += 1
        ratio_step_sizes = step_size / solver_state.step_size
        raw = eng.trial_dual(step_size, solver_state.primal_weight,
                             ratio_step_sizes)
        solver_state.cumulative_kkt_passes += 0.5
        norm_delta_dual_product = math.sqrt(raw[3])
        norm_delta_dual = math.sqrt(raw[2])
        if step_size * norm_delta_dual_product <= \
                step_params.breaking_factor * norm_delta_dual:
            if eng.average_info()[0] == 0:
                eng.add_current_primal_to_average(step_size * ratio_step_sizes)
            eng.accept(solver_state.step_size)
            done = True
        else:
            step_size *= step_params.downscaling_factor
    if it == max_iter and not done:
        solver_state.numerical_error = True
        return
    solver_state.step_size = step_size
    solver_state.ratio_step_sizes = ratio_step_sizes


def take_step(step_params, solver_state, is_lp=True):
    """Dispatch on the policy type like the reference's three methods."""
    if isinstance(step_params, AdaptiveStepsizeParams):
        return take_step_adaptive(step_params, solver_state)
    if isinstance(step_params, MalitskyPockStepsizeParameters):
        return take_step_malitsky_pock(step_params, solver_state, is_lp)
    if isinstance(step_params, ConstantStepsizeParams):
        return take_step_constant(step_params, solver_state)
    raise TypeError(f"unknown step size policy {type(step_params)}")


def take_steps(step_params, solver_state, n_steps, is_lp=True):
    """`n_steps` consecutive take_step calls -- the iterations optimize() runs between
    two termination evaluations (pdhg.jl:862-1046: nothing else happens on them).
    With the HIP engine and the adaptive rule they are one library call
    (pdhg_take_steps_adaptive; the same statements, so the same scalars bit for bit).
    Stops after a step that raised numerical_error.  Returns the steps taken."""
    eng = solver_state.engine
    if (isinstance(step_params, AdaptiveStepsizeParams) and hasattr(eng, "take_steps_adaptive")
            and os.environ.get("PDHG_PY_TAKE_STEP", "0") != "1"):
        (solver_state.step_size, solver_state.total_number_iterations,
         solver_state.cumulative_kkt_passes, err, done) = eng.take_steps_adaptive(
            n_steps, step_params.reduction_exponent, step_params.growth_exponent,
            solver_state.step_size, solver_state.primal_weight,
            solver_state.total_number_iterations, solver_state.cumulative_kkt_passes)
        if err:
            solver_state.numerical_error = True
        return done
    done = 0
    while done < n_steps:
        take_step(step_params, solver_state, is_lp)
        done += 1
        if solver_state.numerical_error:
            break
    return done


# ==============================================================================
# optimize(): the reference's outer loop (pdhg.jl:782-1049) on the host, with
# every n-/m-length vector operation behind ``engine``.
# ==============================================================================
import os
import time as _time

from .evaluation import (POINT_AVERAGE, POINT_CURRENT, DeviceEvaluator,
                         HostEvaluator)
from .iteration_stats_utils import print_to_screen_this_iteration
from .preprocess import rescale_problem, validate
from .quadratic_programming import is_linear_programming_problem
from .saddle_point import (RestartParameters, compute_new_primal_weight,
                           create_last_restart_info, run_restart_scheme,
                           select_initial_primal_weight,
                           unscaled_saddle_point_output,
                           update_objective_bound_estimates)
from .solve_log import PointType, RestartChoice, TerminationReason
from .termination import (TerminationCriteria, cached_quadratic_program_info,
                          check_termination_criteria)


@dataclass
class PdhgParameters:
    """pdhg.jl:128-199 (same field names and order)."""
    l_inf_ruiz_iterations: int
    l2_norm_rescaling: bool
    pock_chambolle_alpha: Optional[float]
    primal_importance: float
    scale_invariant_initial_primal_weight: bool
    verbosity: int
    record_iteration_stats: bool
    termination_evaluation_frequency: int
    termination_criteria: TerminationCriteria
    restart_params: RestartParameters
    step_size_policy_params: object


class EngineOps:
    """A*x, A'*y of the SCALED problem on the device (pdhg_spmv / pdhg_spmv_t);
    Q*x on the host (Q is LP-empty or tiny in the reference's QPs)."""

    def __init__(self, engine, problem):
        self._eng = engine
        self._Q = problem.objective_matrix
        self._n = problem.num_variables

    def Ax(self, x):
        return self._eng.spmv(x)

    def ATy(self, y):
        return self._eng.spmv_t(y)

    def Qx(self, x):
        if self._Q.nnz == 0:
            return np.zeros(self._n)
        return self._Q @ x


class UnscaledEngineOps:
    """The ORIGINAL problem's mat-vecs through the scaled device matrix:
    A = E A_s D  =>  A x = E .* (A_s (D .* x)),  A'y = D .* (A_s' (E .* y))."""

    def __init__(self, engine, scaled_problem):
        self._eng = engine
        self._E = scaled_problem.constraint_rescaling
        self._D = scaled_problem.variable_rescaling
        self._Q = scaled_problem.original_qp.objective_matrix
        self._n = scaled_problem.original_qp.num_variables

    def Ax(self, x):
        return self._E * self._eng.spmv(self._D * x)

    def ATy(self, y):
        return self._D * self._eng.spmv_t(self._E * y)

    def Qx(self, x):
        if self._Q.nnz == 0:
            return np.zeros(self._n)
        return self._Q @ x


def power_method_failure_probability(dimension, epsilon, k):
    """pdhg.jl:378-390"""
    if k < 2 or epsilon <= 0.0:
        return 1.0
    return min(0.824, 0.354 / math.sqrt(epsilon * (k - 1))) * \
        math.sqrt(dimension) * (1.0 - epsilon) ** (k - 1 / 2)


def estimate_maximum_singular_value(ops, num_cols, probability_of_failure=0.01,
                                    desired_relative_error=0.1, seed=1):
    """pdhg.jl:414-440.  The reference draws randn(MersenneTwister(seed)); that
    stream cannot be replayed outside Julia, so parity of the constant-step
    policy is at the tolerance of the power method, not bitwise."""
    epsilon = 1.0 - (1.0 - desired_relative_error) ** 2
    x = np.random.default_rng(seed).standard_normal(num_cols)
    number_of_power_iterations = 0
    while power_method_failure_probability(num_cols, epsilon,
                                           number_of_power_iterations) > probability_of_failure:
        x = x / math.sqrt(float(x @ x))
        x = ops.ATy(ops.Ax(x))
        number_of_power_iterations += 1
    return (math.sqrt(float(x @ ops.ATy(ops.Ax(x))) / float(x @ x)),
            number_of_power_iterations)


def define_norms(primal_size, dual_size, step_size, primal_weight):
    """pdhg.jl:265-277"""
    with np.errstate(divide="ignore"):
        primal_norm_params = np.float64(1) / step_size * primal_weight * np.ones(primal_size)
        dual_norm_params = np.float64(1) / step_size / primal_weight * np.ones(dual_size)
    return primal_norm_params, dual_norm_params


def _default_engine_factory(problem):
    from .engine import HipPdhgEngine
    return HipPdhgEngine.from_problem(problem)


def optimize(params, original_problem, engine_factory=None):
    """``optimize(params::PdhgParameters, original_problem)`` -- pdhg.jl:782-1049.

    ``engine_factory(scaled_qp) -> engine`` builds the device state; the
    default is the HIP engine on the current GPU and there is no CPU fallback.
    Returns a ``SaddlePointOutput``.  The engine (device memory) is released on
    every exit path, exceptions included."""
    created = []
    try:
        return _optimize(params, original_problem, engine_factory, created)
    finally:
        for eng in created:
            if hasattr(eng, "close"):
                eng.close()


def _optimize(params, original_problem, engine_factory, created):
    validate(original_problem)
    qp_cache = cached_quadratic_program_info(original_problem)
    if params.primal_importance <= 0 or not math.isfinite(params.primal_importance):
        raise ValueError("primal_importance must be positive and finite")
    is_lp_original = is_linear_programming_problem(original_problem)
    engine = None
    device_rescale = engine_factory is None or getattr(engine_factory, "takes_original_problem", False)
    if device_rescale and os.environ.get("PDHG_HOST_RESCALE", "0") != "1":
        # Product path: upload the ORIGINAL problem and rescale on the device
        # (pdhg_rescale); only the n-/m-length vectors come back.  The scaled
        # constraint (and objective) matrix lives on the device only.
        from .quadratic_programming import QuadraticProgrammingProblem, ScaledQpProblem
        import scipy.sparse as _sp
        engine = (engine_factory or _default_engine_factory)(original_problem)
        created.append(engine)
        constraint_rescaling, variable_rescaling = engine.rescale(
            params.l_inf_ruiz_iterations, params.l2_norm_rescaling, params.pock_chambolle_alpha)
        c_s, b_s, lb_s, ub_s = engine.get_problem_vectors()
        m0, n0 = original_problem.constraint_matrix.shape
        problem = QuadraticProgrammingProblem(
            lb_s, ub_s, _sp.csc_matrix((n0, n0)), c_s, original_problem.objective_constant,
            _sp.csc_matrix((m0, n0)),      # the scaled matrix lives on the device only
            b_s, original_problem.num_equalities)
        scaled_problem = ScaledQpProblem(original_problem, problem, constraint_rescaling,
                                         variable_rescaling)
        matrix_max_abs = engine.matrix_max_abs()
    else:
        scaled_problem = rescale_problem(params.l_inf_ruiz_iterations,
                                         params.l2_norm_rescaling,
                                         params.pock_chambolle_alpha,
                                         params.verbosity, original_problem)
        problem = scaled_problem.scaled_qp
        data = problem.constraint_matrix.data
        matrix_max_abs = float(np.max(np.abs(data))) if len(data) else 0.0   # norm(A, Inf) on a sparse matrix
    primal_size = problem.num_variables
    dual_size = problem.num_constraints

    if engine is None:
        engine = (engine_factory or _default_engine_factory)(problem)
        created.append(engine)
    ops = EngineOps(engine, problem)
    original_ops = UnscaledEngineOps(engine, scaled_problem)
    is_lp = is_lp_original       # (the host copy of a device-rescaled problem carries no matrices)
    solver_state = PdhgSolverState(engine)   # zeros(...) state, pdhg.jl:805-819
    policy = params.step_size_policy_params

    def inv_max_abs():
        return math.inf if matrix_max_abs == 0.0 else 1.0 / matrix_max_abs

    if isinstance(policy, AdaptiveStepsizeParams):
        solver_state.cumulative_kkt_passes += 0.5
        solver_state.step_size = inv_max_abs()
    elif isinstance(policy, MalitskyPockStepsizeParameters):
        solver_state.cumulative_kkt_passes += 0.5
        solver_state.step_size = inv_max_abs()
        solver_state.ratio_step_sizes = 1.0
    else:
        desired_relative_error = 0.2
        maximum_singular_value, number_of_power_iterations = \
            estimate_maximum_singular_value(ops, primal_size,
                                            probability_of_failure=0.001,
                                            desired_relative_error=desired_relative_error)
        solver_state.step_size = (1 - desired_relative_error) / maximum_singular_value
        solver_state.cumulative_kkt_passes += number_of_power_iterations

    KKT_PASSES_PER_TERMINATION_EVALUATION = 2.0

    if params.scale_invariant_initial_primal_weight:
        solver_state.primal_weight = select_initial_primal_weight(
            problem, np.ones(primal_size), np.ones(dual_size),
            params.primal_importance, params.verbosity)
    else:
        solver_state.primal_weight = params.primal_importance

    primal_weight_update_smoothing = params.restart_params.primal_weight_update_smoothing
    iteration_stats = []
    start_time = _time.time()
    time_spent_doing_basic_algorithm = 0.0

    if getattr(engine, "supports_device_evaluation", False):
        ev = DeviceEvaluator(engine, scaled_problem, qp_cache)
    else:
        ev = HostEvaluator(engine, scaled_problem, qp_cache, ops, original_ops)
    last_restart_info = create_last_restart_info()

    termination_criteria = params.termination_criteria
    iteration_limit = termination_criteria.iteration_limit
    termination_evaluation_frequency = params.termination_evaluation_frequency
    solver_state.numerical_error = False

    iteration = 0
    while True:
        iteration += 1
        if ((iteration - 1) % termination_evaluation_frequency == 0 or
                iteration == iteration_limit + 1 or iteration <= 10 or
                solver_state.numerical_error):
            solver_state.cumulative_kkt_passes += KKT_PASSES_PER_TERMINATION_EVALUATION
            count_x, count_y, _, _ = engine.average_info()
            if solver_state.numerical_error or count_x == 0 or count_y == 0:
                avg_point = POINT_CURRENT
            else:
                avg_point = POINT_AVERAGE

            current_iteration_stats = ev.iteration_stats(
                avg_point, termination_criteria, params.record_iteration_stats, iteration,
                _time.time() - start_time, solver_state.cumulative_kkt_passes,
                solver_state.step_size, solver_state.primal_weight,
                PointType.POINT_TYPE_AVERAGE_ITERATE)
            method_specific_stats = current_iteration_stats.method_specific_stats
            method_specific_stats["time_spent_doing_basic_algorithm"] = \
                time_spent_doing_basic_algorithm

            # define_norms (pdhg.jl:265-277): uniform weights, kept as scalars
            with np.errstate(divide="ignore"):
                primal_weight_norm = float(np.float64(1) / solver_state.step_size * solver_state.primal_weight)
                dual_weight_norm = float(np.float64(1) / solver_state.step_size / solver_state.primal_weight)
            termination_reason = check_termination_criteria(
                termination_criteria, qp_cache, current_iteration_stats)
            if solver_state.numerical_error and termination_reason is False:
                termination_reason = TerminationReason.TERMINATION_REASON_NUMERICAL_ERROR
            # update_objective_bound_estimates (pdhg.jl:938-945) fills three entries of method_specific_stats that are
            # only ever read from KEPT stats (solve_log; the final log, saddle_point.jl:961-993) -- neither the
            # termination test nor the restart scheme sees them.  A check whose stats are dropped skips the two
            # trust-region problems behind them (a quarter of a check on medium LPs); the values of kept stats are
            # the reference's (both functions only read the state, so their order does not matter).
            if params.record_iteration_stats or termination_reason is not False:
                update_objective_bound_estimates(
                    current_iteration_stats.method_specific_stats, ev, avg_point,
                    primal_weight_norm, dual_weight_norm)

            if params.record_iteration_stats or termination_reason is not False:
                iteration_stats.append(current_iteration_stats)

            if print_to_screen_this_iteration(termination_reason, iteration, params.verbosity,
                                              termination_evaluation_frequency):
                _display_iteration_stats(current_iteration_stats)

            if termination_reason is not False:
                # ** Terminate the algorithm ** (the only exit, pdhg.jl:973-992)
                if params.verbosity >= 2:
                    print(f"Terminated after {iteration - 1} iterations: "
                          f"{termination_reason.name}")
                avg_primal_solution, avg_dual_solution = ev.solution(avg_point)
                out = unscaled_saddle_point_output(
                    scaled_problem, avg_primal_solution, avg_dual_solution,
                    termination_reason, iteration - 1, iteration_stats)
                return out

            current_iteration_stats.restart_used = run_restart_scheme(
                ev, last_restart_info, iteration - 1, primal_weight_norm,
                dual_weight_norm, solver_state.primal_weight, params.verbosity,
                params.restart_params)

            if current_iteration_stats.restart_used != RestartChoice.RESTART_CHOICE_NO_RESTART:
                solver_state.primal_weight = compute_new_primal_weight(
                    last_restart_info, solver_state.primal_weight,
                    primal_weight_update_smoothing, params.verbosity)
                solver_state.ratio_step_sizes = 1.0
            # RESTART_TO_AVERAGE: A'y was recomputed inside
            # engine.restart_to_average() (pdhg.jl:1018-1022).

        # This iteration's take_step and those of the iterations up to (not including)
        # the next one the test above fires on: the reference does nothing else on them.
        next_evaluation = ((iteration - 1) // termination_evaluation_frequency + 1) * \
            termination_evaluation_frequency + 1
        if iteration < 10:
            next_evaluation = iteration + 1
        if iteration < iteration_limit + 1:
            next_evaluation = min(next_evaluation, iteration_limit + 1)
        batch = next_evaluation - iteration
        time_spent_doing_basic_algorithm_checkpoint = _time.time()
        iteration += take_steps(policy, solver_state, batch, is_lp) - 1
        time_spent_doing_basic_algorithm += \
            _time.time() - time_spent_doing_basic_algorithm_checkpoint


def _display_iteration_stats(stats):
    """Condensed form of display_iteration_stats (iteration_stats_utils.jl:559-619)."""
    ci = stats.convergence_information[0]
    print("  %6d %9.1f %8.2f | %9.2e %9.2e %9.2e | %12.5e %12.5e | %9.2e %9.2e" % (
        stats.iteration_number, stats.cumulative_kkt_matrix_passes,
        stats.cumulative_time_sec, ci.relative_l2_primal_residual,
        ci.relative_l2_dual_residual, ci.relative_optimality_gap,
        ci.primal_objective, ci.dual_objective, stats.step_size,
        stats.primal_weight))
