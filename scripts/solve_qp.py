#!/usr/bin/env python3
"""Command-line driver with the interface of the reference's scripts/solve_qp.jl
(same flags and defaults, same four output files), running PDHG on the MI355X
through the C ABI.

    python scripts/solve_qp.py --instance_path test.mps --output_dir out --method pdhg

There is no CPU fallback: without a GPU the run fails loudly.  Tests exercise the
same code path on CPU by passing ``engine_factory`` to ``main`` (tests/test_qp_io.py).
"""
import argparse
import gzip
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _bool(s):
    return str(s).lower() in ("1", "true", "t", "yes")


def parse_command_line(argv=None):
    """scripts/solve_qp.jl:170-477 (same names, types and defaults)."""
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--method", required=True)
    ap.add_argument("--output_dir", required=True)
    ap.add_argument("--instance_path", required=True)
    ap.add_argument("--l_inf_ruiz_iterations", type=int, default=10)
    ap.add_argument("--l2_norm_rescaling", type=_bool, default=False)
    ap.add_argument("--pock_chambolle_rescaling", type=_bool, default=True)
    ap.add_argument("--pock_chambolle_alpha", type=float, default=1.0)
    ap.add_argument("--primal_importance", type=float, default=1.0)
    ap.add_argument("--scale_invariant_initial_primal_weight", type=_bool, default=True)
    ap.add_argument("--artificial_restart_threshold", type=float, default=0.5)
    ap.add_argument("--sufficient_reduction_for_restart", type=float, default=0.1)
    ap.add_argument("--necessary_reduction_for_restart", type=float, default=0.9)
    ap.add_argument("--primal_weight_update_smoothing", type=float, default=0.5)
    ap.add_argument("--verbosity", type=int, default=2)
    ap.add_argument("--redirect_stdio", type=_bool, default=False)
    ap.add_argument("--diagonal_scaling", default="off")
    ap.add_argument("--restart_scheme", default="adaptive_normalized")
    ap.add_argument("--restart_frequency", type=int, default=1000)
    ap.add_argument("--restart_to_current_metric", default="gap_over_distance_squared")
    ap.add_argument("--use_approximate_localized_duality_gap", type=_bool, default=False)
    ap.add_argument("--record_iteration_stats", type=_bool, default=True)
    ap.add_argument("--termination_evaluation_frequency", type=int, default=40)
    ap.add_argument("--optimality_norm", default=None)
    ap.add_argument("--absolute_optimality_tol", type=float, default=None)
    ap.add_argument("--relative_optimality_tol", type=float, default=None)
    ap.add_argument("--eps_primal_infeasible", type=float, default=None)
    ap.add_argument("--eps_dual_infeasible", type=float, default=None)
    ap.add_argument("--time_sec_limit", type=float, default=None)
    ap.add_argument("--iteration_limit", type=int, default=None)
    ap.add_argument("--kkt_matrix_pass_limit", type=float, default=None)
    ap.add_argument("--transform_bounds_into_linear_constraints", type=_bool, default=False)
    ap.add_argument("--fixed_format_input", type=_bool, default=False)
    ap.add_argument("--step_size_policy", default="adaptive")
    ap.add_argument("--adaptive_step_size_reduction_exponent", type=float, default=0.3)
    ap.add_argument("--adaptive_step_size_growth_exponent", type=float, default=0.6)
    ap.add_argument("--malitsky_pock_downscaling_factor", type=float, default=0.7)
    ap.add_argument("--malitsky_pock_breaking_factor", type=float, default=0.99)
    ap.add_argument("--malitsky_pock_interpolation_coefficient", type=float, default=1.0)
    return ap.parse_args(argv)


def build_parameters(a):
    """scripts/solve_qp.jl:510-596"""
    import folp_loader
    folp_loader.load()
    from firstorderlp_jl_amd import primal_dual_hybrid_gradient as pdhg
    from firstorderlp_jl_amd import saddle_point as spt
    from firstorderlp_jl_amd import termination as term
    if a.method == "mirror-prox":
        raise SystemExit("mirror-prox is outside this build's scope (PDHG hot path only)")
    if a.method != "pdhg":
        raise SystemExit("`method` arg must be either `mirror-prox` or `pdhg`.")
    schemes = {"no_restart": spt.RestartScheme.NO_RESTARTS,
               "adaptive_normalized": spt.RestartScheme.ADAPTIVE_NORMALIZED,
               "adaptive_distance": spt.RestartScheme.ADAPTIVE_DISTANCE,
               "adaptive_localized": spt.RestartScheme.ADAPTIVE_LOCALIZED,
               "fixed_frequency": spt.RestartScheme.FIXED_FREQUENCY}
    metrics = {"no_restart_to_current": spt.RestartToCurrentMetric.NO_RESTART_TO_CURRENT,
               "gap_over_distance": spt.RestartToCurrentMetric.GAP_OVER_DISTANCE,
               "gap_over_distance_squared": spt.RestartToCurrentMetric.GAP_OVER_DISTANCE_SQUARED}
    if a.restart_scheme not in schemes:
        raise SystemExit(f"Unknown restart scheme {a.restart_scheme}")
    if a.restart_to_current_metric not in metrics:
        raise SystemExit(f"Unknown value for restart_to_current_metric {a.restart_to_current_metric}")
    restart_params = spt.construct_restart_parameters(
        schemes[a.restart_scheme], metrics[a.restart_to_current_metric],
        a.restart_frequency, a.artificial_restart_threshold,
        a.sufficient_reduction_for_restart, a.necessary_reduction_for_restart,
        a.primal_weight_update_smoothing, a.use_approximate_localized_duality_gap)
    alpha = a.pock_chambolle_alpha if a.pock_chambolle_rescaling else None
    tc = term.construct_termination_criteria()
    if a.optimality_norm == "l2":
        tc.optimality_norm = term.L2
    elif a.optimality_norm == "l_inf":
        tc.optimality_norm = term.L_INF
    elif a.optimality_norm is not None:
        raise SystemExit("Unknown termination norm.")
    for field, arg in (("eps_optimal_absolute", "absolute_optimality_tol"),
                       ("eps_optimal_relative", "relative_optimality_tol"),
                       ("eps_primal_infeasible", "eps_primal_infeasible"),
                       ("eps_dual_infeasible", "eps_dual_infeasible"),
                       ("time_sec_limit", "time_sec_limit"),
                       ("iteration_limit", "iteration_limit"),
                       ("kkt_matrix_pass_limit", "kkt_matrix_pass_limit")):
        if getattr(a, arg) is not None:
            setattr(tc, field, getattr(a, arg))
    if a.step_size_policy == "malitsky-pock":
        policy = pdhg.MalitskyPockStepsizeParameters(
            a.malitsky_pock_downscaling_factor, a.malitsky_pock_breaking_factor,
            a.malitsky_pock_interpolation_coefficient)
    elif a.step_size_policy == "constant":
        policy = pdhg.ConstantStepsizeParams()
    else:
        policy = pdhg.AdaptiveStepsizeParams(a.adaptive_step_size_reduction_exponent,
                                             a.adaptive_step_size_growth_exponent)
    return pdhg.PdhgParameters(a.l_inf_ruiz_iterations, a.l2_norm_rescaling, alpha,
                               a.primal_importance, a.scale_invariant_initial_primal_weight,
                               a.verbosity, a.record_iteration_stats,
                               a.termination_evaluation_frequency, tc, restart_params, policy)


def write_vector_to_file(filename, vector):
    """scripts/solve_qp.jl:42-48: one value per line."""
    with open(filename, "w") as fh:
        for x in vector:
            fh.write(repr(float(x)) + "\n")


def solve_instance_and_output(parameters, a, argv, engine_factory=None):
    """scripts/solve_qp.jl:65-162"""
    from firstorderlp_jl_amd import primal_dual_hybrid_gradient as pdhg
    from firstorderlp_jl_amd.preprocess import presolve, undo_presolve
    from firstorderlp_jl_amd.quadratic_programming_io import qps_reader_to_standard_form
    from firstorderlp_jl_amd.solve_log import PointType, SolveLog, to_jsonable
    os.makedirs(a.output_dir, exist_ok=True)
    base = os.path.basename(a.instance_path)
    instance_name = re.sub(r"\.(mps|MPS|qps|QPS)(\.gz)?$", "", base)
    if not re.search(r"\.(mps|qps)(\.gz)?$", base.lower()):
        raise SystemExit(f"Instance has unrecognized file extension: {base}")
    lp = qps_reader_to_standard_form(a.instance_path, fixed_format=a.fixed_format_input)
    presolve_info = presolve(lp, verbosity=parameters.verbosity,
                             transform_bounds=a.transform_bounds_into_linear_constraints)
    if parameters.verbosity >= 1:
        print("Instance: ", instance_name)
    t0 = time.time()
    output = pdhg.optimize(parameters, lp, engine_factory)
    running_time = time.time() - t0
    print(f"Elapsed time: {running_time} sec")

    log = SolveLog()
    log.instance_name = instance_name
    log.command_line_invocation = " ".join([sys.argv[0]] + list(argv))
    log.termination_reason = output.termination_reason
    log.termination_string = output.termination_string
    log.iteration_count = output.iteration_count
    log.solve_time_sec = running_time
    log.solution_stats = output.iteration_stats[-1]
    log.solution_type = PointType.POINT_TYPE_AVERAGE_ITERATE
    with open(os.path.join(a.output_dir, instance_name + "_summary.json"), "w") as fh:
        json.dump(to_jsonable(log), fh)
    log.iteration_stats = output.iteration_stats
    with gzip.open(os.path.join(a.output_dir, instance_name + "_full_log.json.gz"), "wt") as fh:
        json.dump(to_jsonable(log), fh)
    primal, dual = undo_presolve(presolve_info, output.primal_solution, output.dual_solution)
    write_vector_to_file(os.path.join(a.output_dir, instance_name + "_primal.txt"), primal)
    write_vector_to_file(os.path.join(a.output_dir, instance_name + "_dual.txt"), dual)
    return output, primal, dual


def main(argv=None, engine_factory=None):
    """``engine_factory`` is a test hook (see primal_dual_hybrid_gradient.optimize);
    the command line always uses the HIP engine."""
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse_command_line(argv)
    parameters = build_parameters(a)
    return solve_instance_and_output(parameters, a, argv, engine_factory)


if __name__ == "__main__":
    main()
