#!/usr/bin/env python3
"""Kernel-time summary of a `rocprofv3 --kernel-trace --stats` run as JSON.

    python tools/rocprof_summary.py gpurun_out/prof > profiles/rNN_rocprof_summary.json

Walks the directory for rocprofv3 ``*.db`` files and reads the ``top_kernels``
view (name, calls, total / average duration, share)."""
import glob
import json
import os
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


if __name__ == "__main__":
    print(json.dumps({"kernels": summarize(sys.argv[1])}, indent=1))
