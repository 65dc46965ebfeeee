"""bench.py's driver-facing line (round 5's line grew to 22.9 KB and the driver's record of it was
`"parsed": null`).  The line must be ONE compact JSON object of at most 4 KB, the last line of stdout, parseable from the
last 8 KB of stdout, and carry `roofline` and `cpu_baseline`; the whole record lives in bench_details.json.  Replayed on
round 5's own 22.9 KB record (profiles/r05_bench_default.json) through the same final-print path (`bench.py --replay`)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORD = os.path.join(ROOT, "profiles", "r05_bench_default.json")


def _replay(path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--replay", path], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stdout


def test_compact_line_is_last_short_and_parseable():
    out = _replay(RECORD)
    lines = out.decode().rstrip("\n").split("\n")
    line = lines[-1]
    assert len(line.encode()) <= 4096
    d = json.loads(out[-8192:].decode().strip().split("\n")[-1])
    assert d == json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "details"):
        assert k in d, k
    assert d["config"]["workload"] and d["config"]["nnz"] == 100_000_000
    rf = d["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_rocprof", "kernel_ms_rocprof", "avg_launch_ms",
              "algorithmic_bytes_per_launch", "traffic", "traffic_over_algorithmic", "vendor_spmv_ms",
              "rocprof_measured_in_this_run"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "effective_GBps"):
        assert k in cb, k
    assert len(d["other_configs"]) == 2 and all("value" in r and "frac" in r for r in d["other_configs"])


def test_compact_line_of_the_whole_record_keeps_the_headline_numbers():
    import bench
    full = json.load(open(RECORD))
    d = json.loads(bench.compact_line(full))
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"]
    assert d["roofline"]["frac"] == full["roofline"]["frac"]
    assert d["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]


@pytest.mark.parametrize("blow", ["kernel", "workload", "others"])
def test_compact_line_never_exceeds_the_limit(blow):
    """Whatever grows -- a slab product's kernel list, the workload string, more legs -- the line stays under 4 KB."""
    import bench
    full = json.load(open(RECORD))
    if blow == "kernel":
        full["roofline"]["kernel"] = " + ".join(["spmv_stream_kernel<1, true, 0>"] * 400)
    elif blow == "workload":
        full["config"]["workload"] = "x" * 20000
    else:
        full["other_configs"] = full["other_configs"] * 40
    line = bench.compact_line(full)
    assert len(line.encode()) <= bench.COMPACT_LIMIT
    d = json.loads(line)
    assert d["roofline"]["frac"] == full["roofline"]["frac"] and d["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]


def test_multi_gpu_error_record_is_a_compact_line_too():
    import bench
    d = json.loads(bench.compact_line({"metric": "pdhg_iterations_per_sec", "value": None, "n_gpus": 8, "config": {"workload": "random"},
                                       "error": "rank 3: RuntimeError('x')" * 100}))
    assert d["value"] is None and d["n_gpus"] == 8 and len(d["error"]) <= 400
