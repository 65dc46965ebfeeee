#!/bin/bash
# full GPU suite on the final build + the three bench lines through the batched call
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/g22
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/g22/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g22/pytest.log
tail -5 gpurun_out/g22/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/g22/smoke.log 2>&1; tail -2 gpurun_out/g22/smoke.log
for w in l1svm pagerank; do
  for mode in "" "--per-step-calls"; do
    timeout 600 python bench.py --workload $w --steps 4000 --warmup 300 --cpu-seconds 0 --no-other-configs $mode 2>/dev/null | tail -1 > gpurun_out/g22/${w}${mode:+_perstep}.json
    python - <<PY
import json; d=json.load(open("gpurun_out/g22/${w}${mode:+_perstep}.json")); print("$w", "$mode", d["value"], d["ms_per_step"], d.get("host_us_per_trial"))
PY
  done
done
timeout 900 python bench.py > gpurun_out/g22/random.json 2>gpurun_out/g22/random.err; tail -c 600 gpurun_out/g22/random.json
