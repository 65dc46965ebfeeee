mkdir -p gpurun_out/r3t
PRE=$PWD/firstorderlp.jl_amd/csrc/variants/libpdhg_predd.so
run() { name=$1; shift
  env "$@" PDHG_COOP_TRACE=1 PDHG_VERBOSE=1 timeout 300 python bench.py --steps 3000 --warmup 200 --no-cpu-baseline --no-other-configs --profile-steps 0 --workload $WL $EXTRA > gpurun_out/r3t/${WL}_$name.json 2> gpurun_out/r3t/${WL}_$name.err
  python -c "
import json; d=json.load(open('gpurun_out/r3t/${WL}_$name.json')); print('$WL $name', d['value'], d['ms_per_step'])"
  grep -A8 "timeline" gpurun_out/r3t/${WL}_$name.err | grep "phase\|published" | cut -c1-110
}
WL=l1svm
run dd A=1
run predd PDHG_HIP_LIB=$PRE
run dd2 A=1
run predd2 PDHG_HIP_LIB=$PRE
WL=random
EXTRA=""
PDHG_HIP_LIB=$PRE timeout 600 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-other-configs --profile-steps 0 > gpurun_out/r3t/configS_predd.json 2>/dev/null
timeout 600 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-other-configs --profile-steps 0 > gpurun_out/r3t/configS_dd.json 2>/dev/null
PDHG_HIP_LIB=$PRE timeout 600 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-other-configs --profile-steps 0 > gpurun_out/r3t/configS_predd2.json 2>/dev/null
timeout 600 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-other-configs --profile-steps 0 > gpurun_out/r3t/configS_dd2.json 2>/dev/null
for f in configS_predd configS_dd configS_predd2 configS_dd2; do python -c "
import json; d=json.load(open('gpurun_out/r3t/$f.json')); print('$f', d['value'], d['ms_per_step'])"; done
