mkdir -p gpurun_out/r3v
timeout 900 python -m pytest tests/test_gpu_exact_sums.py tests/test_gpu_native_take_step.py tests/test_gpu_step_parity.py -x -q -m gpu 2>&1 | tail -4
V=$PWD/firstorderlp.jl_amd/csrc/variants
run() { name=$1; shift
  env "$@" PDHG_COOP_TRACE=1 PDHG_VERBOSE=1 timeout 300 python bench.py --steps 3000 --warmup 200 --no-cpu-baseline --no-other-configs --profile-steps 0 --workload $WL $EXTRA > gpurun_out/r3v/${WL}_$name.json 2> gpurun_out/r3v/${WL}_$name.err
  python -c "
import json; d=json.load(open('gpurun_out/r3v/${WL}_$name.json')); print('$WL $name', d['value'], d['ms_per_step'])"
  grep -A8 "timeline" gpurun_out/r3v/${WL}_$name.err | grep "phase\|published" | cut -c1-110
}
WL=l1svm
run dpp A=1
run predd PDHG_HIP_LIB=$V/libpdhg_predd.so
run dpp2 A=1
run predd2 PDHG_HIP_LIB=$V/libpdhg_predd.so
