#!/usr/bin/env python3
"""End-to-end optimize() with the reference's solve_qp.jl defaults (Ruiz-10 +
Pock-Chambolle rescaling, adaptive steps, adaptive-normalized restarts,
evaluation every 40 iterations) on a generated LP, on the GPU."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import folp_loader
folp_loader.load()
from firstorderlp_jl_amd.generators import pagerank_lp, random_lp, l1_svm_rcv1_like_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgParameters, optimize
from firstorderlp_jl_amd.saddle_point import RestartScheme, RestartToCurrentMetric, construct_restart_parameters
from firstorderlp_jl_amd.termination import construct_termination_criteria

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="random")
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--tol", type=float, default=1e-4)
ap.add_argument("--iteration_limit", type=int, default=20000)
ap.add_argument("--verbosity", type=int, default=2)
ap.add_argument("--no-record", action="store_true", help="record_iteration_stats = false: only the terminating check's stats are kept")
ap.add_argument("--breakdown", action="store_true", help="time the stages of the evaluation branch (wall clock around the host calls)")
a = ap.parse_args()
stage_time, stage_calls = {}, {}
if a.breakdown:
    import firstorderlp_jl_amd.primal_dual_hybrid_gradient as _pd
    def _timed(mod, name):
        f = getattr(mod, name)
        def g(*args, **kw):
            t = time.perf_counter()
            try:
                return f(*args, **kw)
            finally:
                stage_time[name] = stage_time.get(name, 0.0) + time.perf_counter() - t
                stage_calls[name] = stage_calls.get(name, 0) + 1
        setattr(mod, name, g)
    for nm in ("update_objective_bound_estimates", "check_termination_criteria", "run_restart_scheme",
               "compute_new_primal_weight", "take_steps"):
        _timed(_pd, nm)
    _it = _pd.DeviceEvaluator.iteration_stats
    def _its(self, *args, **kw):
        t = time.perf_counter()
        try:
            return _it(self, *args, **kw)
        finally:
            stage_time["iteration_stats"] = stage_time.get("iteration_stats", 0.0) + time.perf_counter() - t
            stage_calls["iteration_stats"] = stage_calls.get("iteration_stats", 0) + 1
    _pd.DeviceEvaluator.iteration_stats = _its
t0 = time.time()
if a.workload == "pagerank":
    p = pagerank_lp(a.n)
elif a.workload == "l1svm":
    p = l1_svm_rcv1_like_lp()
else:
    p = random_lp(a.n, a.n, 10, 12345)
print(f"generated {a.workload} m={p.num_constraints} n={p.num_variables} nnz={p.constraint_matrix.nnz} in {time.time()-t0:.1f}s", flush=True)
tc = construct_termination_criteria(eps_optimal_absolute=a.tol, eps_optimal_relative=a.tol,
                                    iteration_limit=a.iteration_limit)
rp = construct_restart_parameters(RestartScheme.ADAPTIVE_NORMALIZED, RestartToCurrentMetric.GAP_OVER_DISTANCE_SQUARED,
                                  1000, 0.5, 0.1, 0.9, 0.5, False)
params = PdhgParameters(10, False, 1.0, 1.0, True, a.verbosity, not a.no_record, 40, tc, rp, AdaptiveStepsizeParams(0.3, 0.6))
t0 = time.time()
out = optimize(params, p)
dt = time.time() - t0
last = out.iteration_stats[-1]
ci = last.convergence_information[0]
print(f"{out.termination_string} after {out.iteration_count} iterations in {dt:.2f}s "
      f"(basic algorithm {last.method_specific_stats['time_spent_doing_basic_algorithm']:.2f}s, "
      f"kkt passes {last.cumulative_kkt_matrix_passes:.0f}) "
      f"pobj={ci.primal_objective:.8g} dobj={ci.dual_objective:.8g} "
      f"rel_res=({ci.relative_l2_primal_residual:.2e},{ci.relative_l2_dual_residual:.2e}) gap={ci.relative_optimality_gap:.2e}")
if a.breakdown:
    for k in sorted(stage_time, key=stage_time.get, reverse=True):
        print(f"  {k:36s} {stage_time[k]*1e3:9.2f} ms in {stage_calls[k]:6d} calls = {stage_time[k]/stage_calls[k]*1e6:8.1f} us each")
    print(f"  {'everything else (setup, rescaling, first launches)':36s} {(dt - sum(stage_time.values()))*1e3:9.2f} ms")
