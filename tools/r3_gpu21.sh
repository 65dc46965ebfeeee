mkdir -p gpurun_out/r3x
V=$PWD/firstorderlp.jl_amd/csrc/variants
run() { name=$1; shift
  env "$@" PDHG_COOP_TRACE=1 PDHG_VERBOSE=1 timeout 300 python bench.py --steps 3000 --warmup 200 --no-cpu-baseline --no-other-configs --profile-steps 0 --workload $WL $EXTRA > gpurun_out/r3x/${WL}_$name.json 2> gpurun_out/r3x/${WL}_$name.err
  python -c "
import json; d=json.load(open('gpurun_out/r3x/${WL}_$name.json')); print('$WL $name', d['value'], d['ms_per_step'])"
  grep -A8 "timeline" gpurun_out/r3x/${WL}_$name.err | grep "phase 2\|published" | cut -c1-110
}
WL=l1svm
run b5 A=1
run b4 PDHG_HIP_LIB=$V/libpdhg_b4.so
run predd PDHG_HIP_LIB=$V/libpdhg_predd.so
run b5_again A=1
run b4_again PDHG_HIP_LIB=$V/libpdhg_b4.so
