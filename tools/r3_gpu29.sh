#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "trust or eval or restart or kat or dist_group or lazy or edge or solve or optim or rescale" 2>&1 | grep -v "^[A-Z][A-Za-z]* \(version\|path\) *:\|Hostname" | tail -4
timeout 300 python tools/eval_cost.py l1svm 2>&1 | grep -v amdgpu.ids | grep "point=0\|eval_point\|dist"
timeout 300 python tools/eval_cost.py 1000000 2>&1 | grep -v amdgpu.ids | grep "point=0\|eval_point\|dist"
bash tools/r3_gpu28.sh > /dev/null 2>&1
python - <<'PY'
import csv
for arg in ("l1svm","1000000"):
    print("==",arg)
    for r in csv.DictReader(open(f"gpurun_out/g28_{arg}/eval_kernel_stats.csv")):
        n=r["Name"]
        if any(k in n for k in ("tr_probe","multi_final","tr_setup","eval_cols","eval_rows","dist2")):
            print(n.split("::")[1][:24], r["Calls"], r["AverageNs"])
PY
for w in l1svm pagerank; do timeout 600 python tools/solve_demo.py --workload $w --verbosity 0 --iteration_limit 40000 2>/dev/null | tail -1; done
timeout 600 python tools/solve_demo.py --workload random --n 1000000 --verbosity 0 --iteration_limit 40000 2>/dev/null | tail -1
