set -x
mkdir -p gpurun_out/r3a
export PDHG_VERBOSE=1
python -m pytest tests/test_gpu_dist_group.py tests/test_gpu_big_nnz.py tests/test_gpu_native_take_step.py tests/test_gpu_multi_device.py tests/test_gpu_abi_errors.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r3a/tests.log
unset PDHG_VERBOSE
for t in 1 0; do
  PDHG_SHARD_THREADS=$t python bench.py --shards 8 --steps 100 --warmup 10 --no-other-configs --no-cpu-baseline --profile-steps 0 > gpurun_out/r3a/bench_shards8_threads$t.json 2> gpurun_out/r3a/bench_shards8_threads$t.err
  PDHG_SHARD_THREADS=$t python bench.py --shards 2 --steps 100 --warmup 10 --no-other-configs --no-cpu-baseline --profile-steps 0 > gpurun_out/r3a/bench_shards2_threads$t.json 2> gpurun_out/r3a/bench_shards2_threads$t.err
done
PDHG_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r3a/bench_dist_world1.json 2> gpurun_out/r3a/bench_dist_world1.err
tail -3 gpurun_out/r3a/*.err
cat gpurun_out/r3a/tests.log
