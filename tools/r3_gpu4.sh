mkdir -p gpurun_out/r3e
timeout 900 python -m pytest tests/test_gpu_native_take_step.py tests/test_gpu_row_order.py tests/test_gpu_slabs.py tests/test_gpu_step_parity.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r3e/tests.log
cat gpurun_out/r3e/tests.log
run() { # name, env...
  name=$1; shift
  env "$@" PDHG_COOP_TRACE=1 PDHG_VERBOSE=1 PDHG_SPMV=stream timeout 300 python bench.py --steps 2000 --warmup 100 --no-cpu-baseline --no-other-configs --profile-steps 0 --workload ${WL} > gpurun_out/r3e/bench_${WL}_$name.json 2> gpurun_out/r3e/bench_${WL}_$name.err
  python -c "
import json; d=json.load(open('gpurun_out/r3e/bench_${WL}_$name.json')); print('${WL} $name', d['value'], d['ms_per_step'], d['layout']['trial_graph'], d.get('host_us_per_trial'))"
  grep -A8 "timeline" gpurun_out/r3e/bench_${WL}_$name.err | cut -c1-120
}
WL=l1svm
run w5 A=1
run w8 PDHG_HIP_LIB=$PWD/firstorderlp.jl_amd/csrc/variants/libpdhg_w8.so
run w5_512 PDHG_COOP_WGS=512
WL=pagerank
run graph_slabs PDHG_COOP=0
run w5 PDHG_SLABS=0
run w8 PDHG_SLABS=0 PDHG_HIP_LIB=$PWD/firstorderlp.jl_amd/csrc/variants/libpdhg_w8.so
