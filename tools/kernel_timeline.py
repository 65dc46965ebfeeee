#!/usr/bin/env python3
"""Dev tool: per-iteration GPU timeline from a `rocprofv3 --kernel-trace` database:
kernel start/end relative to the first kernel of the window, and the idle gaps.
usage: kernel_timeline.py DIR [first_dispatch] [count]"""
import glob, os, sqlite3, sys
root = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
db = sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True))[0]
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("PRAGMA table_info(kernels)")]
rows = list(con.execute("SELECT * FROM kernels ORDER BY start"))
recs = [dict(zip(cols, r)) for r in rows]
win = recs[first:first + count]
t0 = win[0]["start"]
prev_end = None
for r in win:
    gap = (r["start"] - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(r['start']-t0)/1e3:9.2f} us  dur {(r['end']-r['start'])/1e3:7.2f}  gap {gap:7.2f}  {r['name'][:70]}")
    prev_end = max(prev_end or 0, r["end"])
busy = sum(r["end"] - r["start"] for r in win) / 1e3
print(f"window {(win[-1]['end']-t0)/1e3:.1f} us, kernel-busy {busy:.1f} us")
