# dev: the default bench command, per-kernel event brackets printed (is the roofline's bracket sane?)
mkdir -p gpurun_out/r5g
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-vendor > gpurun_out/r5g/bench_a.json 2> gpurun_out/r5g/bench_a.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5g/bench_a.json"))
print(d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"].get("kernel_ms_rocprof"))
for k, v in d["kernels"].items():
    print(k[:40], v["avg_ms"], v["launches"])
PY
