# round 6: row data requested a trip ahead in the stream kernels' row phase (stream_row_request)
export PDHG_DEV=1 SHAPE_CACHE_DIR=/tmp/shapes
mkdir -p gpurun_out/r6 /tmp/shapes
T=gpurun_out/r6/row_prefetch.txt
: > $T
for wl in l1svm pagerank; do for rep in 1 2; do
  python bench.py --workload $wl --steps 4000 --warmup 300 --no-cpu-baseline --no-self-profile --no-vendor --no-details 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('bench $wl', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" >> $T
done; done
python bench.py --workload random --rows 100000 --cols 100000 --steps 3000 --warmup 300 --no-cpu-baseline --no-self-profile --no-vendor --no-details --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('random 100K', d['value'], d['ms_per_step'], d['launch_path'])" >> $T
python tools/shape_table.py --only "l1svm,pagerank 1M,200K-uniform,200K-banded,200K-lognormal" --no-vendor 2>&1 | grep -v "^#" | cut -c1-220 >> $T
cat $T
timeout 1500 python -m pytest tests/test_gpu_stream_pipe.py tests/test_gpu_step_parity.py tests/test_gpu_edge_shapes.py tests/test_gpu_slabs.py tests/test_gpu_row_order.py tests/test_gpu_device_loop.py tests/test_gpu_sj.py tests/test_gpu_full_size.py -x -q 2>&1 | tail -4
