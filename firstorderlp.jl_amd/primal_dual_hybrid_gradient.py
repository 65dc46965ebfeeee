"""Host driver of the PDHG inner loop: the reference's
src/primal_dual_hybrid_gradient.jl with the n-/m-length vector work moved
behind the C ABI (``engine.HipPdhgEngine``).

What stays on the host (as the reference's north star prescribes): the scalar
step-size rules of ``take_step`` (pdhg.jl:555-767), counters, and -- in
``optimize`` -- restarts, primal-weight updates and termination.
"""
import math
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


def take_step_adaptive(step_params, solver_state):
    """take_step(::AdaptiveStepsizeParams, ...)  pdhg.jl:653-731."""
    eng = solver_state.engine
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
        step_size = min(first_term, second_term)
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
