#!/bin/bash
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
# small LPs: one launch per trial against the batch of steps in one workgroup with the vectors in LDS (small_lp_kernel.hpp)
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_small_lp.py -x -q 2>&1 | grep -v "^[A-Z][A-Za-z]* \(version\|path\) *:\|Hostname" | tail -5
run() { python bench.py "${@:2}" --steps 4000 --warmup 300 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config']['nnz'])"; }
for shape in "30 30 3" "300 300 5" "500 500 8" "1000 800 6"; do
  set -- $shape
  PDHG_SMALL_FEW_ROWS=0 run "1024 threads m=$1 n=$2" --m $1 --n $2 --nnz-per-row $3
  PDHG_SMALL_FEW_ROWS=100000 run "256 threads  m=$1 n=$2" --m $1 --n $2 --nnz-per-row $3
done
