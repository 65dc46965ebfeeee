#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { python bench.py "${@:2}" --steps 4000 --warmup 300 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('host_us_per_trial'))"; }
for rep in 1 2; do
  PDHG_DEVICE_LOOP=0 run "loop0 l1svm" --workload l1svm
  PDHG_DEVICE_LOOP=1 run "loop1 l1svm" --workload l1svm
  PDHG_DEVICE_LOOP=0 run "loop0 r100k" --m 100000 --n 100000
  PDHG_DEVICE_LOOP=1 run "loop1 r100k" --m 100000 --n 100000
  PDHG_DEVICE_LOOP=0 run "loop0 r250k" --m 250000 --n 250000
  PDHG_DEVICE_LOOP=1 run "loop1 r250k" --m 250000 --n 250000
done
