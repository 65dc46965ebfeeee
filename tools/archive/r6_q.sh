export PDHG_DEV=1
timeout 1500 python -m pytest tests/test_gpu_fake_rccl.py -x -q -k "all_gather or overlapped" 2>&1 | tail -12
export HSA_ENABLE_IPC_MODE_LEGACY=0
PDHG_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29811 bench.py --gpus 8 --steps 20 --warmup 5 --dist-overlap > gpurun_out/r6prof/r06_bench_fake_rccl_8ranks_overlap_line.json 2> gpurun_out/r6prof/bench_fake8.err
echo rc=$?; cp bench_details.json gpurun_out/r6prof/r06_bench_fake_rccl_8ranks_overlap.json; tail -1 gpurun_out/r6prof/r06_bench_fake_rccl_8ranks_overlap_line.json | cut -c1-400
