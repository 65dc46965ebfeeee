"""Output schema mirroring src/solve_log.jl (same enum members and field
names, so ``*_summary.json`` files are comparable with the reference's)."""
import enum
from dataclasses import dataclass, field, asdict
from typing import Dict, List


class RestartChoice(enum.Enum):
    """solve_log.jl:32-37"""
    RESTART_CHOICE_UNSPECIFIED = 0
    RESTART_CHOICE_NO_RESTART = 1
    RESTART_CHOICE_WEIGHTED_AVERAGE_RESET = 2
    RESTART_CHOICE_RESTART_TO_AVERAGE = 3


class PointType(enum.Enum):
    """solve_log.jl:52-58"""
    POINT_TYPE_UNSPECIFIED = 0
    POINT_TYPE_CURRENT_ITERATE = 1
    POINT_TYPE_ITERATE_DIFFERENCE = 2
    POINT_TYPE_AVERAGE_ITERATE = 3
    POINT_TYPE_NONE = 4


class TerminationReason(enum.Enum):
    """solve_log.jl:336-347"""
    TERMINATION_REASON_UNSPECIFIED = 0
    TERMINATION_REASON_OPTIMAL = 1
    TERMINATION_REASON_PRIMAL_INFEASIBLE = 2
    TERMINATION_REASON_DUAL_INFEASIBLE = 3
    TERMINATION_REASON_TIME_LIMIT = 4
    TERMINATION_REASON_ITERATION_LIMIT = 5
    TERMINATION_REASON_KKT_MATRIX_PASS_LIMIT = 6
    TERMINATION_REASON_NUMERICAL_ERROR = 7
    TERMINATION_REASON_INVALID_PROBLEM = 8
    TERMINATION_REASON_OTHER = 9


# module-level aliases so call sites read like the reference
for _e in (RestartChoice, PointType, TerminationReason):
    for _m in _e:
        globals()[_m.name] = _m


@dataclass
class ConvergenceInformation:
    """solve_log.jl:64-172"""
    candidate_type: PointType = PointType.POINT_TYPE_UNSPECIFIED
    primal_objective: float = 0.0
    dual_objective: float = 0.0
    corrected_dual_objective: float = 0.0
    l_inf_primal_residual: float = 0.0
    l2_primal_residual: float = 0.0
    l_inf_dual_residual: float = 0.0
    l2_dual_residual: float = 0.0
    relative_l_inf_primal_residual: float = 0.0
    relative_l2_primal_residual: float = 0.0
    relative_l_inf_dual_residual: float = 0.0
    relative_l2_dual_residual: float = 0.0
    relative_optimality_gap: float = 0.0
    l_inf_primal_variable: float = 0.0
    l2_primal_variable: float = 0.0
    l_inf_dual_variable: float = 0.0
    l2_dual_variable: float = 0.0


@dataclass
class InfeasibilityInformation:
    """solve_log.jl:174-230"""
    candidate_type: PointType = PointType.POINT_TYPE_UNSPECIFIED
    max_primal_ray_infeasibility: float = 0.0
    primal_ray_linear_objective: float = 0.0
    primal_ray_quadratic_norm: float = 0.0
    max_dual_ray_infeasibility: float = 0.0
    dual_ray_objective: float = 0.0


@dataclass
class IterationStats:
    """solve_log.jl:232-334"""
    iteration_number: int = 0
    convergence_information: List[ConvergenceInformation] = field(default_factory=list)
    infeasibility_information: List[InfeasibilityInformation] = field(default_factory=list)
    cumulative_kkt_matrix_passes: float = 0.0
    cumulative_rejected_steps: int = 0
    cumulative_time_sec: float = 0.0
    restart_used: RestartChoice = RestartChoice.RESTART_CHOICE_UNSPECIFIED
    step_size: float = 0.0
    primal_weight: float = 0.0
    method_specific_stats: Dict[str, float] = field(default_factory=dict)


@dataclass
class SolveLog:
    """solve_log.jl:349-420"""
    instance_name: str = ""
    command_line_invocation: str = ""
    termination_reason: TerminationReason = TerminationReason.TERMINATION_REASON_UNSPECIFIED
    termination_string: str = ""
    iteration_count: int = 0
    solve_time_sec: float = 0.0
    solution_stats: IterationStats = field(default_factory=IterationStats)
    solution_type: PointType = PointType.POINT_TYPE_UNSPECIFIED
    iteration_stats: List[IterationStats] = field(default_factory=list)


def to_jsonable(obj):
    """JSON3/StructTypes.Mutable equivalent: enums by name, dataclasses by field."""
    if isinstance(obj, enum.Enum):
        return obj.name
    if hasattr(obj, "__dataclass_fields__"):
        return {k: to_jsonable(getattr(obj, k)) for k in obj.__dataclass_fields__}
    if isinstance(obj, (list, tuple)):
        return [to_jsonable(v) for v in obj]
    if isinstance(obj, dict):
        return {k: to_jsonable(v) for k, v in obj.items()}
    if isinstance(obj, float):
        import math
        if math.isinf(obj) or math.isnan(obj):
            return None
    return obj
