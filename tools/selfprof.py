#!/usr/bin/env python3
"""bench.py measures its own roofline evidence: it runs ITSELF (a short child run of the same workload) under
`rocprofv3 --kernel-trace --stats` and under the two `--pmc` passes MI355X_MICROARCH.md prescribes (counter passes
apart from each other and from any other trace domain), then reads the kernel durations and the fabric bytes of the
dominant product out of rocprofv3's databases.  The figures in the bench line's `roofline` object are then produced
inside the driver's own run instead of being quoted from files under profiles/.

Used by bench.py (import) and runnable by hand:

    python tools/selfprof.py --workload random "spmv_tiled_kernel<1, 0>"

Formulae (as tools/pmc_traffic.sh, calibrated on the elementwise kernels whose byte counts are known):
    read bytes  = 32 * RDREQ_32B + 128 * RDREQ_128B + 64 * (RDREQ - RDREQ_32B - RDREQ_128B)
    write bytes = 1024 * WRITE_SIZE
per launch, summed over the kernels of the product (its launches per product taken from the trace).
"""
import glob
import json
import os
import re
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMC_PASSES = (("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_HIT_sum"),
              ("WRITE_SIZE", "TCC_MISS_sum"))


def short_name(full):
    """'void (anonymous namespace)::spmv_stream_kernel<1, true, 0>(args...)' -> 'spmv_stream_kernel<1, true, 0>'"""
    m = re.search(r"([A-Za-z_0-9]+(?:<[^>]*>)?)\(", full)
    return m.group(1) if m else full


def kernel_times(root):
    """{short kernel name: (calls, average microseconds)} from the top_kernels view of every database under root."""
    out = {}
    for db in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        con = sqlite3.connect(db)
        try:
            cols = [r[1] for r in con.execute("PRAGMA table_info(top_kernels)")]
            if not cols:
                continue
            for rec in con.execute("SELECT * FROM top_kernels"):
                d = dict(zip(cols, rec))
                out[short_name(d["name"])] = (int(d["total_calls"]), float(d["average"]))
        finally:
            con.close()
    return out


def counters(root):
    """{short kernel name: {counter: (dispatches, mean value)}} from the counters_collection view."""
    out = {}
    for db in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        con = sqlite3.connect(db)
        try:
            cols = [r[1] for r in con.execute("PRAGMA table_info(counters_collection)")]
            if not cols:
                continue
            kcol = "kernel_name" if "kernel_name" in cols else "name"
            for k, c, n, v in con.execute(f"SELECT {kcol}, counter_name, COUNT(*), AVG(value) FROM counters_collection "
                                          f"GROUP BY {kcol}, counter_name"):
                out.setdefault(short_name(k), {})[c] = (int(n), float(v))
        finally:
            con.close()
    return out


def product_time_us(times, product):
    """Sum over the kernels of one fused product (names joined by ' + ') of average duration x launches per product."""
    members = [m.strip() for m in product.split("+")]
    if not all(m in times for m in members):
        return None
    carriers = [times[m][0] for m in members if "long" not in m] or [times[members[0]][0]]
    per_product = min(carriers)
    return sum(times[m][1] * times[m][0] / per_product for m in members)


def product_traffic_bytes(ctr, product):
    members = [m.strip() for m in product.split("+")]
    if not all(m in ctr and "TCC_EA0_RDREQ_sum" in ctr[m] and "WRITE_SIZE" in ctr[m] for m in members):
        return None, None
    detail = {}
    for m in members:
        c = ctr[m]
        r, r32, r128 = c["TCC_EA0_RDREQ_sum"][1], c.get("TCC_EA0_RDREQ_32B_sum", (0, 0.0))[1], c.get("TCC_EA0_RDREQ_128B_sum", (0, 0.0))[1]
        reads = 32 * r32 + 128 * r128 + 64 * (r - r32 - r128)
        writes = 1024.0 * c["WRITE_SIZE"][1]
        detail[m] = {"launches": c["TCC_EA0_RDREQ_sum"][0], "read_bytes": int(reads), "write_bytes": int(writes),
                     "l2_hits": int(c.get("TCC_HIT_sum", (0, 0.0))[1]), "l2_misses": int(c.get("TCC_MISS_sum", (0, 0.0))[1])}
    carriers = [detail[m]["launches"] for m in members if "long" not in m] or [detail[members[0]]["launches"]]
    base = min(carriers)
    total = sum((d["read_bytes"] + d["write_bytes"]) * d["launches"] / base for d in detail.values())
    for d in detail.values():      # L2 requests (hits + misses) of one product's launches of this kernel
        d["l2_requests_per_product"] = int((d["l2_hits"] + d["l2_misses"]) * d["launches"] / base)
    return int(total), detail


def child_command(workload, extra):
    # the separate-launch path (PDHG_GRAPH=0 in the child's environment): the kernels are the ones the one-launch paths
    # run, launched one by one so that durations and counters attribute to them
    return [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "4", "--warmup", "2",
            "--no-cpu-baseline", "--no-other-configs", "--profile-steps", "0", "--no-self-profile", "--no-vendor", "--no-details"] + list(extra)


def run(workload, product, extra=(), timeout=180, keep=None, deadline=None):
    """Profile a short child run of bench.py for `workload`; `product` is the dominant product's kernel list
    (pdhg_kernel_name).  Returns a dict (kernel_ms, traffic, detail, seconds) or {"error": ...}; never raises."""
    rp = shutil.which("rocprofv3")
    if not rp:
        return {"error": "rocprofv3 not on PATH"}
    t0 = time.time()
    work = keep or tempfile.mkdtemp(prefix="pdhg_selfprof_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", PDHG_GRAPH="0")
    out = {"command": " ".join(child_command(workload, extra)[1:]) + " (child run, PDHG_GRAPH=0: every kernel its own launch)"}
    try:
        passes = [("kt", ["--kernel-trace", "--stats"])] + [(f"p{i + 1}", ["--kernel-trace", "--pmc"] + list(c)) for i, c in enumerate(PMC_PASSES)]
        for tag, flags in passes:
            if deadline is not None and time.time() > deadline:      # the bench line must not wait on its evidence
                out.setdefault("failed_passes", []).append({"pass": tag, "rc": None, "stderr_tail": "skipped: time budget spent"})
                continue
            d = os.path.join(work, tag)
            cmd = [rp] + flags + ["-d", d, "--"] + child_command(workload, extra)
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            except subprocess.TimeoutExpired:
                out.setdefault("failed_passes", []).append({"pass": tag, "rc": None, "stderr_tail": f"timed out after {timeout} s"})
                continue
            if r.returncode != 0:
                out.setdefault("failed_passes", []).append({"pass": tag, "rc": r.returncode, "stderr_tail": r.stderr.decode(errors="replace")[-400:]})
        times = kernel_times(os.path.join(work, "kt"))
        us = product_time_us(times, product)
        if us is not None:
            out["kernel_ms"] = round(us * 1e-3, 5)
            out["kernels"] = {m.strip(): {"calls": times[m.strip()][0], "avg_us": round(times[m.strip()][1], 2)} for m in product.split("+")}
        ctr = {}
        for i in range(len(PMC_PASSES)):
            for k, v in counters(os.path.join(work, f"p{i + 1}")).items():
                ctr.setdefault(k, {}).update(v)
        traffic, detail = product_traffic_bytes(ctr, product)
        if traffic is not None:
            out["traffic"] = traffic
            out["traffic_detail"] = detail
            out["l2_requests"] = sum(d["l2_requests_per_product"] for d in detail.values())
    except Exception as exc:      # measurement extra: the bench line must not depend on it
        out["error"] = repr(exc)
    finally:
        if keep is None:
            shutil.rmtree(work, ignore_errors=True)
    out["seconds"] = round(time.time() - t0, 1)
    return out


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="random")
    ap.add_argument("--keep", default=None, help="directory to keep rocprofv3's output in")
    ap.add_argument("product")
    ap.add_argument("extra", nargs="*")
    a = ap.parse_args()
    print(json.dumps(run(a.workload, a.product, a.extra, keep=a.keep), indent=1))
