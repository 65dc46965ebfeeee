mkdir -p gpurun_out/r3j
timeout 600 python -m pytest tests/test_gpu_native_take_step.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r3j/tests.log
cat gpurun_out/r3j/tests.log
PDHG_VERBOSE=1 timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-other-configs --profile-steps 0 > gpurun_out/r3j/create_configS.json 2> gpurun_out/r3j/create_configS.err
grep -i "pdhg_create\|tiled layout" gpurun_out/r3j/create_configS.err
nproc; lscpu | grep -i "model name\|^CPU(s)\|Thread\|Socket"
bash tools/r3_profile.sh > gpurun_out/r3j/profile.log 2>&1
tail -12 gpurun_out/r3j/profile.log
