export PDHG_DEV=1 SHAPE_CACHE_DIR=/tmp/shapes
mkdir -p gpurun_out/r6 /tmp/shapes
T=gpurun_out/r6/pipe_small.txt
: > $T
for env in "" "PDHG_STREAM_PIPE=0"; do
  echo "## env: $env" >> $T
  env $env python tools/shape_table.py --only "l1svm,200K-uniform,200K-banded,200K-lognormal" --no-vendor 2>&1 | grep -v "^#" | cut -c1-220 >> $T
  for wl in l1svm; do
    env $env python bench.py --workload $wl --steps 4000 --warmup 300 --no-cpu-baseline --no-self-profile --no-vendor --no-details 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'][:80])" >> $T
  done
  env $env timeout 600 python tools/solve_demo.py --workload l1svm --iteration_limit 40000 --verbosity 0 --breakdown 2>&1 | tail -7 | head -4 >> $T
  env $env PDHG_COOP=0 python bench.py --workload random --rows 100000 --cols 100000 --steps 2000 --warmup 200 --no-cpu-baseline --no-self-profile --no-vendor --no-details --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('random 100K graph path', d['value'], d['ms_per_step'], d['launch_path'])" >> $T
done
cat $T
timeout 1200 python -m pytest tests/test_gpu_stream_pipe.py tests/test_gpu_step_parity.py tests/test_gpu_edge_shapes.py tests/test_gpu_lazy_accept.py tests/test_gpu_device_eval.py -x -q 2>&1 | tail -5
