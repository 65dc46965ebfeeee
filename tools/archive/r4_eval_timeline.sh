#!/bin/bash
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
# one termination / restart check of an L1-SVM solve on the GPU timeline: the kernels between two steps_kernel launches
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
W=${1:-l1svm}
rm -rf gpurun_out/r4evtl; mkdir -p gpurun_out/r4evtl
rocprofv3 --kernel-trace -d gpurun_out/r4evtl/kt -- python tools/solve_demo.py --workload $W --iteration_limit 2000 --verbosity 0 > /dev/null 2>&1
python - "$W" <<'PY'
import glob, sqlite3, sys
db = sorted(glob.glob("gpurun_out/r4evtl/kt/**/*.db", recursive=True))[0]
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("PRAGMA table_info(kernels)")]
recs = [dict(zip(cols, r)) for r in con.execute("SELECT * FROM kernels ORDER BY start")]
idx = [i for i, r in enumerate(recs) if "steps_kernel" in r["name"] or "trial_kernel" in r["name"] or "hipGraph" in r["name"]]
# the window between the 30th and the 31st batch of steps
big = [i for i in idx if recs[i]["end"] - recs[i]["start"] > 300e3]
a, b = big[30], big[31]
t0 = recs[a]["end"]; prev = t0
print(f"# {sys.argv[1]}: kernels between two batches of steps (after the steps kernel that ended at t=0; next one starts at the end)")
busy = 0
for r in recs[a + 1:b + 1]:
    print(f"{(r['start']-t0)/1e3:9.2f} us  dur {(r['end']-r['start'])/1e3:7.2f}  gap {(r['start']-prev)/1e3:7.2f}  {r['name'].replace('(anonymous namespace)::','')[:64]}")
    prev = max(prev, r["end"]); busy += r["end"] - r["start"]
busy -= recs[b]["end"] - recs[b]["start"]
print(f"# check window {(recs[b]['start']-t0)/1e3:.1f} us, kernels busy {busy/1e3:.1f} us, launches {b-a-1}")
PY
