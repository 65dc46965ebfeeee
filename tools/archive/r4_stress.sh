#!/bin/bash
# repeated runs of the one-launch / multi-step tests, and long bitwise comparisons of the multi-step kernel with one launch per trial
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r4stress
fail=0
for i in $(seq 1 4); do
  timeout 900 python -m pytest tests/test_gpu_native_take_step.py tests/test_gpu_lazy_accept.py tests/test_gpu_row_order.py tests/test_gpu_device_loop.py tests/test_gpu_small_lp.py tests/test_gpu_trust_region.py tests/test_gpu_dist_group.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r4stress/run$i.log 2>&1 || { fail=$((fail+1)); echo "run $i FAILED"; tail -20 gpurun_out/r4stress/run$i.log; }
done
echo "stress: $fail failures of 4 runs"
tail -1 gpurun_out/r4stress/run4.log
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import folp_loader; pkg = folp_loader.load()
import numpy as np
from firstorderlp_jl_amd.generators import l1_svm_rcv1_like_lp, random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_steps
from tests import helpers as H
bad = 0
for name, p, batches in (("l1svm", l1_svm_rcv1_like_lp(), [64] * 150), ("random 100K", random_lp(100000, 100000, 10, 3), [64] * 100),
                         ("random 30K", random_lp(30000, 25000, 8, 4), [17, 64, 200] * 30),
                         ("random 4K (XCD-local mode)", random_lp(4000, 3500, 8, 5), [17, 64, 200] * 60)):
    outs = []
    for loop in ("0", "1"):
        os.environ["PDHG_DEVICE_LOOP"] = loop
        eng = pkg.HipPdhgEngine.from_problem(p)
        step, pw = H.initial_step_and_weight(p)
        st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
        for k in batches:
            take_steps(AdaptiveStepsizeParams(0.3, 0.6), st, k)
        outs.append(eng.get_current() + eng.get_average() + (np.array([st.step_size, st.total_number_iterations]),))
        eng.close()
    same = all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))
    bad += not same
    print(f"{name}: {sum(batches)} steps, {int(outs[0][4][1])} trials, multi-step kernel == one launch per trial: {same}")
print("long comparisons:", "all bitwise equal" if not bad else f"{bad} MISMATCHES")
PY
python - <<'PY'
# the trust-region search as one persistent launch, many calls in a row: every call must return what the pass-by-pass form returns
import os, sys
sys.path.insert(0, os.getcwd())
import folp_loader; pkg = folp_loader.load()
import numpy as np
from firstorderlp_jl_amd.generators import l1_svm_rcv1_like_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_steps
from tests import helpers as H
p = l1_svm_rcv1_like_lp()
res = {}
for coop in ("0", "1"):
    os.environ["PDHG_TR_COOP"] = coop
    eng = pkg.HipPdhgEngine.from_problem(p)
    step, pw = H.initial_step_and_weight(p)
    st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
    take_steps(AdaptiveStepsizeParams(0.3, 0.6), st, 50); eng.save_restart_point()
    out = []
    for it in range(400):
        take_steps(AdaptiveStepsizeParams(0.3, 0.6), st, 3)
        for point, rng, radius in ((1, 1, 1.0), (1, 2, 1.0), (0, 0, 0.7), (1, 0, 2.5)):
            out.append(eng.trust_region_bound(point, 1.3, 0.8, radius, rng, False))
    res[coop] = np.array(out); calls = eng.layout_info()["tr_coop_calls"]; eng.close()
scale = np.maximum(np.abs(res["0"]).max(axis=0), 1e-300)
dev = np.abs(res["1"][:, :6] - res["0"][:, :6]) / scale[:6]
print(f"trust-region: {len(res['1'])} calls as one persistent launch ({calls} counted), max deviation from pass-by-pass {dev.max():.2e} of the column scale, pass counts equal: {np.array_equal(res['1'][:, 6], res['0'][:, 6])}")
PY
