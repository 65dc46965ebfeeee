mkdir -p gpurun_out/r3q
timeout 1200 python -m pytest tests/test_gpu_exact_sums.py -x -q -m gpu 2>&1 | tail -25
timeout 900 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r3q/bench.json 2> gpurun_out/r3q/bench.err
python -c "
import json
d=json.load(open('gpurun_out/r3q/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v['avg_ms'] for k,v in d['kernels'].items()})
for o in d.get('other_configs', []): print(o['config']['workload'][:40], o.get('value'))
"
