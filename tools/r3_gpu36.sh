#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_device_loop.py -x -q 2>&1 | grep -v "^[A-Z][A-Za-z]* \(version\|path\) *:\|Hostname" | tail -5
PDHG_COOP_TRACE=1 PDHG_DEVICE_LOOP=1 python bench.py --workload l1svm --steps 2000 --warmup 300 --no-cpu-baseline --no-other-configs 2>&1 | grep -A7 "timeline" | cut -c1-200 | grep -v '^{'
run() { python bench.py "${@:2}" --steps 4000 --warmup 300 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  PDHG_DEVICE_LOOP=0 run "loop0 l1svm" --workload l1svm
  PDHG_DEVICE_LOOP=1 run "loop1 l1svm" --workload l1svm
  PDHG_DEVICE_LOOP=0 run "loop0 r100k" --m 100000 --n 100000
  PDHG_DEVICE_LOOP=1 run "loop1 r100k" --m 100000 --n 100000
done
