#!/bin/bash
# variants: libpdhg_{A,B,C,D}.so built with -DPDHG_STEPS_NOINLINE={0,0,1,1} -DPDHG_STEPS_PREFETCH={1,0,1,0} into firstorderlp.jl_amd/csrc/variants/ (git-ignored)
cd "$GRAFT_REPO_ROOT"
run() { python bench.py "${@:2}" --steps 4000 --warmup 300 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for v in A B C D; do
  export PDHG_HIP_LIB=firstorderlp.jl_amd/csrc/variants/libpdhg_$v.so
  PDHG_DEVICE_LOOP=1 run "$v loop1 l1svm" --workload l1svm
  PDHG_DEVICE_LOOP=1 run "$v loop1 r100k" --m 100000 --n 100000
  PDHG_COOP_TRACE=1 PDHG_DEVICE_LOOP=1 python bench.py --workload l1svm --steps 2000 --warmup 300 --no-cpu-baseline --no-other-configs 2>&1 | grep -A7 "timeline" | cut -c1-200 | grep -v '^{'
done
PDHG_DEVICE_LOOP=0 run "A loop0 l1svm" --workload l1svm
