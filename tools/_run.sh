mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.log 2>&1; grep -n "passed\|failed\|error" gpurun_out/gpu_suite.log | tail -5; grep -n "^FAILED\|^E  " gpurun_out/gpu_suite.log | head
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'])
for o in d.get('other_configs',[]): print("   ", o.get('value'), o.get('ms_per_step'), o.get('error'))
PY
