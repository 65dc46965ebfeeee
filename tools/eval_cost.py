#!/usr/bin/env python3
"""Dev tool: what one termination/restart check costs on config S, call by call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import folp_loader
pkg = folp_loader.load()
from firstorderlp_jl_amd.generators import random_lp, l1_svm_rcv1_like_lp, pagerank_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_step
arg = sys.argv[1] if len(sys.argv) > 1 else "10000000"
if arg == "l1svm":
    p = l1_svm_rcv1_like_lp()
elif arg == "pagerank":
    p = pagerank_lp(1_000_000)
else:
    p = random_lp(int(arg), int(arg), 10, 12345)
eng = pkg.HipPdhgEngine.from_problem(p)
m, n = p.num_constraints, p.num_variables
eng.set_original_problem(np.ones(m), np.ones(n), p.objective_vector, p.right_hand_side,
                         p.variable_lower_bound, p.variable_upper_bound)
st = PdhgSolverState(eng, step_size=1.0 / np.abs(p.constraint_matrix.data).max(), primal_weight=1.0)
for _ in range(60):
    take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
eng.save_restart_point()
for _ in range(40):
    take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
def t(name, f, reps=3):
    best = 1e9
    for _ in range(reps):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), st)      # invalidates the cached products
        t0 = time.perf_counter(); r = f(); best = min(best, time.perf_counter() - t0)
    print(f"{name:44s} {best*1e3:8.3f} ms", flush=True)
    return r
t("take_step (for scale)", lambda: take_step(AdaptiveStepsizeParams(0.3, 0.6), st))
t("eval_point(AVERAGE) (fresh products)", lambda: eng.eval_point(1))
t("eval_point(AVERAGE) again (cached) x1", lambda: (eng.eval_point(1), eng.eval_point(1))[1])
t("point_sumsq(AVERAGE)", lambda: eng.point_sumsq(1))
t("distance_to_restart(AVERAGE)", lambda: eng.distance_to_restart(1))
def tr(point, rng):
    eng.eval_point(point)
    t0 = time.perf_counter(); o = eng.trust_region_bound(point, 2.0, 0.5, 0.7, rng); return (time.perf_counter() - t0, o[6])
for point in (1, 0, 2):
    for rng in (0, 1, 2):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
        dt, passes = tr(point, rng)
        print(f"trust_region_bound(point={point}, range={rng}) products cached: {dt*1e3:8.3f} ms, {int(passes)} probe passes", flush=True)
