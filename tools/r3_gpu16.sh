mkdir -p gpurun_out/r3r
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r3r/bench.json 2> gpurun_out/r3r/bench.err
python -c "
import json
d=json.load(open('gpurun_out/r3r/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v['avg_ms'] for k,v in d['kernels'].items()})
for o in d.get('other_configs', []): print(o['config']['workload'][:40], o.get('value'), o.get('ms_per_step'))
"
