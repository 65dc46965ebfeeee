set -x
mkdir -p gpurun_out/r3c
timeout 1500 python -m pytest tests/test_gpu_native_take_step.py tests/test_gpu_row_order.py tests/test_gpu_lazy_accept.py tests/test_gpu_slabs.py tests/test_gpu_step_parity.py tests/test_gpu_tiled.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r3c/tests.log
cat gpurun_out/r3c/tests.log
for wl in l1svm pagerank; do
 for mode in coop graph plain; do
  case $mode in coop) G=1; C=1;; graph) G=1; C=0;; plain) G=0; C=0;; esac
  PDHG_VERBOSE=1 PDHG_GRAPH=$G PDHG_COOP=$C PDHG_SPMV=stream timeout 300 python bench.py --workload $wl --steps 2000 --warmup 100 --no-cpu-baseline --no-other-configs > gpurun_out/r3c/bench_${wl}_${mode}.json 2> gpurun_out/r3c/bench_${wl}_${mode}.err
  python -c "
import json; d=json.load(open('gpurun_out/r3c/bench_${wl}_${mode}.json')); print('$wl $mode', d['value'], d['ms_per_step'], d['layout'], d.get('host_us_per_trial'))"
 done
done
# pagerank variants: no slabs (coop eligible), relaxed sweep
PDHG_SLABS=0 PDHG_SPMV=stream timeout 300 python bench.py --workload pagerank --steps 2000 --warmup 100 --no-cpu-baseline --no-other-configs > gpurun_out/r3c/bench_pagerank_noslab_coop.json 2> gpurun_out/r3c/bench_pagerank_noslab_coop.err
PDHG_VERBOSE=1 timeout 300 python bench.py --workload pagerank --steps 2000 --warmup 100 --no-cpu-baseline --no-other-configs > gpurun_out/r3c/bench_pagerank_auto.json 2> gpurun_out/r3c/bench_pagerank_auto.err
PDHG_ROW_ORDER=strict PDHG_SPMV=stream timeout 300 python bench.py --workload l1svm --steps 2000 --warmup 100 --no-cpu-baseline --no-other-configs > gpurun_out/r3c/bench_l1svm_strict_coop.json 2> gpurun_out/r3c/bench_l1svm_strict_coop.err
for f in pagerank_noslab_coop pagerank_auto l1svm_strict_coop; do python -c "
import json; d=json.load(open('gpurun_out/r3c/bench_$f.json')); print('$f', d['value'], d['ms_per_step'], d['layout'], d.get('host_us_per_trial'), {k:(v['avg_ms']) for k,v in d['kernels'].items()})"; done
grep -h "pdhg_hip\]" gpurun_out/r3c/*.err | sort | uniq -c | head -30
