#!/usr/bin/env python3
"""dev: what one batched trust-region call costs per probe pass on the L1-SVM LP (out[6] = probe passes of each search)."""
import os
import sys
import time
os.environ.setdefault("PDHG_DEV", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import folp_loader
pkg = folp_loader.load()
from firstorderlp_jl_amd.generators import l1_svm_rcv1_like_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_step

p = l1_svm_rcv1_like_lp()
eng = pkg.HipPdhgEngine.from_problem(p)
m, n = p.num_constraints, p.num_variables
eng.set_original_problem(np.ones(m), np.ones(n), p.objective_vector, p.right_hand_side, p.variable_lower_bound, p.variable_upper_bound)
st = PdhgSolverState(eng, step_size=1.0 / np.abs(p.constraint_matrix.data).max(), primal_weight=1.0)
for _ in range(200):
    take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
eng.save_restart_point()
for _ in range(64):
    take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
for radius in (0.1, 1.0, 10.0):
    for pts, rng in (((1, 0, 2), (0, 0, 0)), ((1, 1), (1, 2)), ((1,), (0,))):
        eng.eval_point(1)
        best, rows = 1e9, None
        for _ in range(5):
            t0 = time.perf_counter()
            rows = eng.trust_region_bounds(list(pts), 1.0, 1.0, [radius] * len(pts), list(rng), False)
            best = min(best, time.perf_counter() - t0)
        passes = [int(r[6]) for r in rows]
        print(f"radius {radius:5.1f}  problems {len(pts)}  {best * 1e6:7.1f} us  passes {passes}  -> {best * 1e6 / max(passes):6.1f} us per pass", flush=True)
