#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q -k "trust or eval or restart or kat or dist_group or lazy or edge or solve or optim" 2>&1 | grep -v "^[A-Z][A-Za-z]* \(version\|path\) *:\|Hostname" | tail -5
timeout 300 python tools/eval_cost.py l1svm 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/eval_cost.py 1000000 2>&1 | grep -v amdgpu.ids
for w in l1svm pagerank; do timeout 600 python tools/solve_demo.py --workload $w --verbosity 0 --iteration_limit 40000 2>/dev/null | tail -1; done
timeout 600 python tools/solve_demo.py --workload random --n 1000000 --verbosity 0 --iteration_limit 40000 2>/dev/null | tail -1
