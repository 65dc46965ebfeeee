mkdir -p gpurun_out/r3l
run() { # name, env...
  name=$1; shift
  env "$@" PDHG_COOP_TRACE=1 PDHG_VERBOSE=1 timeout 300 python bench.py --steps 2000 --warmup 100 --no-cpu-baseline --no-other-configs --profile-steps 0 --workload ${WL} $EXTRA > gpurun_out/r3l/bench_${WL}_$name.json 2> gpurun_out/r3l/bench_${WL}_$name.err
  python -c "
import json; d=json.load(open('gpurun_out/r3l/bench_${WL}_$name.json')); print('${WL} $name', d['value'], d['ms_per_step'], d['layout']['trial_graph'], d.get('host_us_per_trial'))"
  grep -A8 "timeline" gpurun_out/r3l/bench_${WL}_$name.err | cut -c1-120
}
W4=$PWD/firstorderlp.jl_amd/csrc/variants/libpdhg_w4.so
WL=l1svm
run w5 A=1
run w4 PDHG_HIP_LIB=$W4
WL=random
EXTRA="--m 100000 --n 100000"
run w5_100k A=1
run w4_100k PDHG_HIP_LIB=$W4
EXTRA="--m 250000 --n 250000"
run w5_250k A=1
run w4_250k PDHG_HIP_LIB=$W4
