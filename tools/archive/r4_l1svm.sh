#!/bin/bash
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
# round 4: the multi-step kernel variants on the latency-bound LPs (run on the GPU box)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_l1svm.txt
: > $OUT
run() { python bench.py "${@:2}" --steps 4000 --warmup 300 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])" >> $OUT; }
trace() { echo "== $1" >> $OUT; PDHG_VERBOSE=1 PDHG_COOP_TRACE=1 python bench.py "${@:2}" --steps 2000 --warmup 300 --no-cpu-baseline --no-other-configs 2>&1 | grep -A7 "timeline\|row blocks filled\|one-launch trial:" | cut -c1-220 | grep -v '^{' >> $OUT; }
for v in ${VARIANTS:-r3 new}; do
  if [ $v = new ]; then unset PDHG_HIP_LIB; else export PDHG_HIP_LIB=firstorderlp.jl_amd/csrc/variants/libpdhg_$v.so; fi
  for bal in ${BALS:-0 1}; do
    [ $v = r3 ] && [ $bal = 1 ] && continue
    export PDHG_BALANCED_BLOCKS=$bal
    run "$v bal$bal l1svm" --workload l1svm
    run "$v bal$bal l1svm(2)" --workload l1svm
    run "$v bal$bal r100k" --m 100000 --n 100000
    run "$v bal$bal r250k" --m 250000 --n 250000
    trace "$v bal$bal l1svm" --workload l1svm
  done
done
unset PDHG_HIP_LIB PDHG_BALANCED_BLOCKS
cat $OUT
