#!/bin/bash
# repeated runs of the one-launch / multi-step tests, and long bitwise comparisons of the multi-step kernel with one launch per trial
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3stress
fail=0
for i in $(seq 1 10); do
  timeout 900 python -m pytest tests/test_gpu_native_take_step.py tests/test_gpu_lazy_accept.py tests/test_gpu_row_order.py tests/test_gpu_device_loop.py tests/test_gpu_small_lp.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r3stress/run$i.log 2>&1 || { fail=$((fail+1)); echo "run $i FAILED"; tail -20 gpurun_out/r3stress/run$i.log; }
done
echo "stress: $fail failures of 10 runs"
tail -1 gpurun_out/r3stress/run10.log
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
                         ("random 30K", random_lp(30000, 25000, 8, 4), [17, 64, 200] * 30)):
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
