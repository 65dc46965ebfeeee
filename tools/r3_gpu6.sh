mkdir -p gpurun_out/r3i
timeout 900 python -m pytest tests/test_gpu_native_take_step.py tests/test_gpu_row_order.py tests/test_gpu_lazy_accept.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r3i/tests.log
cat gpurun_out/r3i/tests.log
run() { # name, env...
  name=$1; shift
  env "$@" PDHG_COOP_TRACE=1 PDHG_VERBOSE=1 timeout 300 python bench.py --steps 2000 --warmup 100 --no-cpu-baseline --no-other-configs --profile-steps 0 --workload ${WL} $EXTRA > gpurun_out/r3i/bench_${WL}_$name.json 2> gpurun_out/r3i/bench_${WL}_$name.err
  python -c "
import json; d=json.load(open('gpurun_out/r3i/bench_${WL}_$name.json')); print('${WL} $name', d['value'], d['ms_per_step'], d['layout']['trial_graph'], d.get('host_us_per_trial'))"
  grep -A8 "timeline" gpurun_out/r3i/bench_${WL}_$name.err | cut -c1-120
}
WL=l1svm
run default A=1
run strict PDHG_ROW_ORDER=strict
WL=random
EXTRA="--m 250000 --n 250000"
run coop_250k A=1
run graph_250k PDHG_COOP=0
EXTRA="--m 100000 --n 100000"
run coop_100k A=1
run graph_100k PDHG_COOP=0
WL=pagerank
EXTRA=""
run default A=1
