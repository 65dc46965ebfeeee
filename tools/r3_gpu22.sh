#!/bin/bash
# bench lines through the batched call vs one call per take_step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/g22
for rep in 1 2; do
for w in l1svm pagerank; do
  for mode in "" "--per-step-calls"; do
    f=gpurun_out/g22/${w}${mode:+_perstep}.json
    timeout 600 python bench.py --workload $w --steps 4000 --warmup 300 --no-cpu-baseline --no-other-configs $mode 2>gpurun_out/g22/err.log | tail -1 > $f
    python - <<PY
import json; d=json.load(open("$f")); print("$w", "$mode", d["value"], d["ms_per_step"], d.get("host_us_per_trial"))
PY
  done
done
done
