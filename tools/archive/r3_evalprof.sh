#!/bin/bash
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
# cProfile of optimize() (solve_qp.jl defaults) on one generated workload: where the time outside take_step goes
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3w
for W in ${WORKLOADS:-l1svm pagerank}; do
W=$W python - <<'PY' > gpurun_out/r3w/evalprof_$W.txt 2>&1
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.getcwd())
import folp_loader; folp_loader.load()
from firstorderlp_jl_amd.generators import l1_svm_rcv1_like_lp, pagerank_lp, random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgParameters, optimize
from firstorderlp_jl_amd.saddle_point import RestartScheme, RestartToCurrentMetric, construct_restart_parameters
from firstorderlp_jl_amd.termination import construct_termination_criteria
w = os.environ["W"]
p = l1_svm_rcv1_like_lp() if w == "l1svm" else (pagerank_lp(1_000_000) if w == "pagerank" else random_lp(1_000_000, 1_000_000, 10, 12345))
tc = construct_termination_criteria(eps_optimal_absolute=1e-4, eps_optimal_relative=1e-4, iteration_limit=40000)
rp = construct_restart_parameters(RestartScheme.ADAPTIVE_NORMALIZED, RestartToCurrentMetric.GAP_OVER_DISTANCE_SQUARED, 1000, 0.5, 0.1, 0.9, 0.5, False)
params = PdhgParameters(10, False, 1.0, 1.0, True, 0, True, 40, tc, rp, AdaptiveStepsizeParams(0.3, 0.6))
optimize(params, p)   # warm
pr = cProfile.Profile(); pr.enable()
t0 = time.time(); out = optimize(params, p); dt = time.time() - t0
pr.disable()
print(w, out.termination_string, out.iteration_count, f"{dt:.3f}s")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue())
PY
head -50 gpurun_out/r3w/evalprof_$W.txt
done
