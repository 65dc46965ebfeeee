mkdir -p gpurun_out/r3w
python - <<'PY' > gpurun_out/r3w/evalprof.txt 2>&1
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.getcwd())
import folp_loader; folp_loader.load()
from firstorderlp_jl_amd.generators import l1_svm_rcv1_like_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgParameters, optimize
from firstorderlp_jl_amd.saddle_point import RestartScheme, RestartToCurrentMetric, construct_restart_parameters
from firstorderlp_jl_amd.termination import construct_termination_criteria
p = l1_svm_rcv1_like_lp()
tc = construct_termination_criteria(eps_optimal_absolute=1e-4, eps_optimal_relative=1e-4, iteration_limit=40000)
rp = construct_restart_parameters(RestartScheme.ADAPTIVE_NORMALIZED, RestartToCurrentMetric.GAP_OVER_DISTANCE_SQUARED, 1000, 0.5, 0.1, 0.9, 0.5, False)
params = PdhgParameters(10, False, 1.0, 1.0, True, 0, True, 40, tc, rp, AdaptiveStepsizeParams(0.3, 0.6))
optimize(params, p)   # warm
pr = cProfile.Profile(); pr.enable()
t0 = time.time(); out = optimize(params, p); dt = time.time() - t0
pr.disable()
print(out.termination_string, out.iteration_count, f"{dt:.3f}s")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35); print(s.getvalue())
PY
head -70 gpurun_out/r3w/evalprof.txt
