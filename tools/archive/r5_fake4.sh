# the driver's N = 4 command over the test-only stand-in transport (one GPU): a functional run of the one-process-per-GPU path
mkdir -p gpurun_out/r5b
export HSA_ENABLE_IPC_MODE_LEGACY=0
PDHG_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29811 bench.py --gpus 4 --steps 20 --warmup 5 > gpurun_out/r5b/bench_fake4_configS.json 2> gpurun_out/r5b/bench_fake4_configS.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5b/bench_fake4_configS.json"))
print(d["value"], d["n_gpus"], d.get("transport"), d["roofline"]["kernel"], d["roofline"]["frac"])
PY
tail -3 gpurun_out/r5b/bench_fake4_configS.err
