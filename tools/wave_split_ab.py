#!/usr/bin/env python3
"""A/B of the shared wave reduction of the evaluation / trust-region kernels (eval_kernels.hpp: WaveSplit) against the
one-tree-per-quantity build (-DPDHG_NO_WAVE_SPLIT, tools/variants.sh): the SAME BITS from every check entry point, and
what a check costs.  Run once per library (PDHG_HIP_LIB); prints a digest of all outputs and the timings.
Usage: python tools/wave_split_ab.py [l1svm|pagerank|<n>]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import folp_loader
pkg = folp_loader.load()
from firstorderlp_jl_amd.generators import random_lp, l1_svm_rcv1_like_lp, pagerank_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_steps

arg = sys.argv[1] if len(sys.argv) > 1 else "l1svm"
p = l1_svm_rcv1_like_lp() if arg == "l1svm" else (pagerank_lp(1_000_000) if arg == "pagerank" else random_lp(int(arg), int(arg), 10, 12345))
eng = pkg.HipPdhgEngine.from_problem(p)
m, n = p.num_constraints, p.num_variables
eng.set_original_problem(np.ones(m), np.ones(n), p.objective_vector, p.right_hand_side, p.variable_lower_bound, p.variable_upper_bound)
st = PdhgSolverState(eng, step_size=1.0 / np.abs(p.constraint_matrix.data).max(), primal_weight=1.0)
take_steps(AdaptiveStepsizeParams(0.3, 0.6), st, 60)
eng.save_restart_point()
take_steps(AdaptiveStepsizeParams(0.3, 0.6), st, 40)
h = hashlib.sha256()
best = {}
for rep in range(9):
    take_steps(AdaptiveStepsizeParams(0.3, 0.6), st, 1)
    t0 = time.perf_counter(); ev = [eng.eval_point(pt) for pt in (1, 0)]; t1 = time.perf_counter()
    b3 = eng.trust_region_bounds([1, 0, 2], 2.0, 0.5, [0.7, 0.7, 0.7], [0, 0, 0]); t2 = time.perf_counter()
    b2 = eng.trust_region_bounds([1, 1], 2.0, 0.5, [0.7, 0.7], [1, 2]); t3 = time.perf_counter()
    singles = [eng.trust_region_bound(pt, 2.0, 0.5, r, rng) for pt in (0, 1) for r in (0.01, 0.7, 50.0) for rng in (0, 1, 2)]; t4 = time.perf_counter()
    d = [eng.distance_to_restart(1), eng.point_sumsq(0)]
    for a in ev + [b3, b2] + singles + [np.array(d)]:
        h.update(np.ascontiguousarray(np.asarray(a, dtype=np.float64)).tobytes())
    for k, v in (("eval_point x2", t1 - t0), ("three bounds, one launch", t2 - t1), ("two bounds, one launch", t3 - t2), ("18 single bounds", t4 - t3)):
        best[k] = min(best.get(k, 1e9), v)
print(arg, "one-launch searches", eng.layout_info()["tr_coop_calls"], "digest", h.hexdigest()[:16], " ".join(f"| {k}: {v * 1e6:.0f} us" for k, v in best.items()))
