mkdir -p gpurun_out/r5b
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
export SHAPE_CACHE_DIR=/tmp/shape_cache
(time python tools/shape_table.py) > gpurun_out/r5b/shape_table_vendor.txt 2> gpurun_out/r5b/shape_table_vendor.err
tail -20 gpurun_out/r5b/shape_table_vendor.txt
bash tools/pmc_stream.sh "banded 10M +-50000" banded50k > gpurun_out/r5b/pmc_banded.log 2>&1
tail -5 gpurun_out/r5b/pmc_banded.log
PDHG_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29811 bench.py --gpus 4 --steps 20 --warmup 5 > gpurun_out/r5b/bench_fake4_configS.json 2> gpurun_out/r5b/bench_fake4_configS.err
tail -c 1500 gpurun_out/r5b/bench_fake4_configS.json; tail -3 gpurun_out/r5b/bench_fake4_configS.err
