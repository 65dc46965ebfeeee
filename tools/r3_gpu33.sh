#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { # label, env...
  python bench.py "${@:2}" --steps 4000 --warmup 300 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"
}
for rep in 1 2 3; do
  PDHG_HIP_LIB=firstorderlp.jl_amd/csrc/variants/libpdhg_prev.so run "prev l1svm" --workload l1svm
  run "new  l1svm" --workload l1svm
  PDHG_HIP_LIB=firstorderlp.jl_amd/csrc/variants/libpdhg_prev.so run "prev pagerank" --workload pagerank
  run "new  pagerank" --workload pagerank
  PDHG_HIP_LIB=firstorderlp.jl_amd/csrc/variants/libpdhg_prev.so run "prev r100k" --m 100000 --n 100000
  run "new  r100k" --m 100000 --n 100000
done
