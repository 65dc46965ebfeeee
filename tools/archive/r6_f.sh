export PDHG_DEV=1
mkdir -p gpurun_out/r6
T=gpurun_out/r6/stream_waitcnt3.txt
: > $T
for wl in pagerank l1svm; do
  echo "## bench --workload $wl (slab passes throttled, the rest not)" >> $T
  for rep in 1 2; do
  python bench.py --workload $wl --steps 4000 --warmup 300 --no-cpu-baseline --no-self-profile --no-vendor --no-details 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" >> $T
  done
done
cat $T
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) 2>&1 | tee gpurun_out/r6/full_gpu_suite.txt
