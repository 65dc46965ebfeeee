#!/usr/bin/env python3
"""Kernel-time summary of a `rocprofv3 --kernel-trace --stats` run as JSON.

    python tools/rocprof_summary.py gpurun_out/prof [bench_line.json] > profiles/rNN_<workload>_rocprof_summary.json

Walks the directory for rocprofv3 ``*.db`` files and reads the ``top_kernels`` view (name,
calls, total / average duration, share).  With the bench line of the SAME command as second
argument it also adds up the kernels of each fused product (the names `pdhg_kernel_name`
reports, joined by " + ": column-slab passes, long-row pair) so that the line's
``roofline`` can be recomputed from this one file:

    frac = algorithmic_bytes / (sum over the group's kernels of avg_us * launches per product) / 8e12

The separate launches live in bench.py's profiling pass (HIP-event bracketed, plain launch
path); the timed region of a small LP runs as ONE `trial_kernel` per trial, listed too."""
import glob
import json
import os
import re
import sqlite3
import sys


def summarize(root):
    rows = []
    for db in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        con = sqlite3.connect(db)
        try:
            cols = [r[1] for r in con.execute("PRAGMA table_info(top_kernels)")]
            if not cols:
                continue
            for rec in con.execute("SELECT * FROM top_kernels"):
                d = dict(zip(cols, rec))
                # rocprofv3's view reports microseconds (total_duration, average)
                rows.append({"name": d["name"], "calls": d["total_calls"],
                             "total_us": round(d["total_duration"], 1),
                             "avg_us": round(d["average"], 2),
                             "pct": round(d["percentage"], 3)})
        finally:
            con.close()
    rows.sort(key=lambda r: -r["total_us"])
    return rows


def short_name(full):
    """'void (anonymous namespace)::spmv_stream_kernel<1, true, 0>(args...)' -> 'spmv_stream_kernel<1, true, 0>'"""
    m = re.search(r"([A-Za-z_0-9]+(?:<[^>]*>)?)\(", full)
    return m.group(1) if m else full


def groups(rows, bench):
    """Per fused product of the bench line: its kernels, their calls / averages, the sum per product."""
    by_name = {}
    for r in rows:
        by_name.setdefault(short_name(r["name"]), []).append(r)
    out = {}
    # (the appended other_configs share kernel names with each other -- one process, three LPs: only the headline
    #  workload's products can be told apart in this trace; configs[2]/[3] have summaries of their own commands)
    lines = [bench]
    for line in lines:
        for label, k in (line.get("kernels") or {}).items():
            if "spmv" not in label:
                continue
            members = [m.strip() for m in label.split("+")]
            found = [(m, by_name[m][0]) for m in members if m in by_name]
            if len(found) != len(members):
                continue
            # launches per product: every member's calls relative to the member that carries the epilogue (the last
            # stream / tiled kernel of the list runs once per product)
            per_product = min(r["calls"] for _, r in found if "long" not in _) if any("long" not in m for m, _ in found) else found[0][1]["calls"]
            total = sum(r["avg_us"] * r["calls"] / per_product for _, r in found)
            entry = {"kernels": {m: {"calls": r["calls"], "avg_us": r["avg_us"]} for m, r in found},
                     "products": per_product, "sum_avg_us_per_product": round(total, 2),
                     "algorithmic_bytes": k["algorithmic_bytes"],
                     "GBps": round(k["algorithmic_bytes"] / (total * 1e-6) / 1e9, 1),
                     "frac_of_8TBps": round(k["algorithmic_bytes"] / (total * 1e-6) / 8e12, 4),
                     "bench_line_event_bracket_ms": k["avg_ms"]}
            out[f"{label} @ {line['config']['workload'][:60]}"] = entry
    return out


if __name__ == "__main__":
    rows = summarize(sys.argv[1])
    doc = {"kernels": rows}
    if len(sys.argv) > 2:
        with open(sys.argv[2]) as fh:
            text = fh.read().strip().splitlines()[-1]
        doc["products"] = groups(rows, json.loads(text))
    print(json.dumps(doc, indent=1))
