mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.log 2>&1; grep -n "passed\|failed\|error" gpurun_out/gpu_suite.log | tail -5
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_lazy.json 2> gpurun_out/bench_lazy.err
PDHG_LAZY_ACCEPT=0 timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_eager.json 2> gpurun_out/bench_eager.err
python - <<'PY'
import json
for f in ("lazy","eager"):
    d=json.loads(open(f"gpurun_out/bench_{f}.json").read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], {k:v['avg_ms'] for k,v in d['kernels'].items()})
    for o in d.get('other_configs',[]): print("   ", o['value'], o['ms_per_step'], {k:v['avg_ms'] for k,v in o['kernels'].items()})
PY
