mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests/test_gpu_row_order.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r3d/tests.log
cat gpurun_out/r3d/tests.log
for ord in relaxed strict; do
PDHG_ROW_ORDER=$ord PDHG_COOP_TRACE=1 PDHG_VERBOSE=1 PDHG_SPMV=stream timeout 300 python bench.py --workload l1svm --steps 2000 --warmup 100 --no-cpu-baseline --no-other-configs --profile-steps 0 > gpurun_out/r3d/bench_l1svm_$ord.json 2> gpurun_out/r3d/bench_l1svm_$ord.err
python -c "
import json; d=json.load(open('gpurun_out/r3d/bench_l1svm_$ord.json')); print('l1svm $ord', d['value'], d['ms_per_step'], d.get('host_us_per_trial'))"
grep -A8 "timeline" gpurun_out/r3d/bench_l1svm_$ord.err
done
PDHG_SLABS=0 PDHG_COOP_TRACE=1 PDHG_VERBOSE=1 PDHG_SPMV=stream timeout 300 python bench.py --workload pagerank --steps 1000 --warmup 100 --no-cpu-baseline --no-other-configs --profile-steps 0 > gpurun_out/r3d/bench_pagerank.json 2> gpurun_out/r3d/bench_pagerank.err
python -c "
import json; d=json.load(open('gpurun_out/r3d/bench_pagerank.json')); print('pagerank noslab coop', d['value'], d['ms_per_step'], d.get('host_us_per_trial'))"
grep -B2 -A8 "timeline" gpurun_out/r3d/bench_pagerank.err
