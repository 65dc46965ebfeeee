mkdir -p gpurun_out/r3s
WL=l1svm
for i in 1 2; do
PDHG_COOP_TRACE=1 PDHG_VERBOSE=1 timeout 300 python bench.py --steps 2000 --warmup 100 --no-cpu-baseline --no-other-configs --profile-steps 0 --workload ${WL} > gpurun_out/r3s/bench_$i.json 2> gpurun_out/r3s/bench_$i.err
python -c "
import json; d=json.load(open('gpurun_out/r3s/bench_$i.json')); print('${WL}', d['value'], d['ms_per_step'], d.get('host_us_per_trial'))"
grep -A8 "timeline" gpurun_out/r3s/bench_$i.err | cut -c1-120
done
