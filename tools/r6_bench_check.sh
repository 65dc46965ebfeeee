# the driver's command, checked the way the driver reads it: the last stdout line is ONE JSON object <= 4 KB that parses
# from the last 8 KB of stdout and carries roofline + cpu_baseline; the whole record is in bench_details.json
mkdir -p gpurun_out/r6
python bench.py "$@" > gpurun_out/r6/bench_stdout.txt 2> gpurun_out/r6/bench_stderr.txt
echo "rc=$?"
cp bench_details.json gpurun_out/r6/bench_details.json 2>/dev/null
python - <<'PY'
import json
raw = open("gpurun_out/r6/bench_stdout.txt", "rb").read()
line = raw[-8192:].decode().rstrip("\n").split("\n")[-1]
d = json.loads(line)
assert len(line.encode()) <= 4096, len(line)
assert raw.decode().rstrip("\n").split("\n")[-1] == line
for k in ("roofline", "cpu_baseline", "value", "ms_per_step", "config"):
    assert d.get(k) is not None, k
print("line bytes", len(line), "stdout lines", raw.count(b"\n"))
print(line)
PY
