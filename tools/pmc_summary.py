#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in rocprofv3 output databases.

    python tools/pmc_summary.py gpurun_out/pmc_sq [substring-of-kernel-name]

Walks the directory for rocprofv3 ``*.db`` files (one per --pmc pass), reads
the ``counters_collection`` view and prints, per kernel and counter, the
number of dispatches and the mean value as JSON.
"""
import glob
import json
import os
import sqlite3
import sys


def summarize(root, needle=None):
    out = {}
    for db in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        con = sqlite3.connect(db)
        try:
            cols = [r[1] for r in con.execute("PRAGMA table_info(counters_collection)")]
            if not cols:
                continue
            kcol = "kernel_name" if "kernel_name" in cols else "name"
            rows = con.execute(
                f"SELECT {kcol}, counter_name, COUNT(*), AVG(value) FROM counters_collection "
                f"GROUP BY {kcol}, counter_name").fetchall()
        finally:
            con.close()
        for kname, cname, cnt, avg in rows:
            if needle and needle not in kname:
                continue
            out.setdefault(kname, {})[cname] = {"dispatches": cnt, "avg": avg}
    return out


if __name__ == "__main__":
    print(json.dumps(summarize(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None), indent=1))
