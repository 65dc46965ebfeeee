export PDHG_DEV=1
mkdir -p gpurun_out/r6
T=gpurun_out/r6/throttle_depth.txt
: > $T
V=$PWD/firstorderlp.jl_amd/csrc/variants
for lib in "" $V/libpdhg_thr2.so $V/libpdhg_thr4.so; do
  echo "## bench --workload pagerank lib=${lib##*/} (throttle: wait after every 1 / 2 / 4 entry-load pairs of a slab pass)" >> $T
  for rep in 1 2 3; do
  PDHG_HIP_LIB=$lib python bench.py --workload pagerank --steps 4000 --warmup 300 --no-cpu-baseline --no-self-profile --no-vendor --no-details 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" >> $T
  done
done
cat $T
( time timeout 1800 python -m pytest tests -x -q -m gpu --durations=45 2>&1 | tail -70 ) 2>&1 | tee gpurun_out/r6/full_gpu_suite.txt
