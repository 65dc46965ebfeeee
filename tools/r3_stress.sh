mkdir -p gpurun_out/r3stress
fail=0
for i in $(seq 1 15); do
  timeout 600 python -m pytest tests/test_gpu_native_take_step.py tests/test_gpu_lazy_accept.py tests/test_gpu_row_order.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r3stress/run$i.log 2>&1 || { fail=$((fail+1)); echo "run $i FAILED"; tail -20 gpurun_out/r3stress/run$i.log; }
done
echo "stress: $fail failures of 15 runs"
tail -2 gpurun_out/r3stress/run15.log
