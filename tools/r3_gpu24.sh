#!/bin/bash
# evaluation-branch round trips: pinned result words vs copy + synchronise
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/g24
for hw in 0 1; do
  echo "== PDHG_EVAL_HOST_WORD=$hw l1svm"
  PDHG_EVAL_HOST_WORD=$hw timeout 300 python tools/eval_cost.py l1svm 2>&1 | grep -v amdgpu.ids
done
for hw in 0 1; do
  echo "== PDHG_EVAL_HOST_WORD=$hw solves"
  for w in l1svm pagerank; do PDHG_EVAL_HOST_WORD=$hw timeout 600 python tools/solve_demo.py --workload $w --verbosity 0 --iteration_limit 40000 2>/dev/null | tail -1; done
  PDHG_EVAL_HOST_WORD=$hw timeout 600 python tools/solve_demo.py --workload random --n 1000000 --verbosity 0 --iteration_limit 40000 2>/dev/null | tail -1
done
timeout 1200 python -m pytest tests -m gpu -x -q -k "eval or trust or restart or solve or kat or optimize" 2>&1 | tail -3
