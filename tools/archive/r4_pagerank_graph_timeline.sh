cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
rm -rf gpurun_out/prtl; mkdir -p gpurun_out/prtl
rocprofv3 --kernel-trace -d gpurun_out/prtl/kt -- python bench.py --workload pagerank --steps 60 --warmup 20 --no-cpu-baseline --no-other-configs --no-self-profile --no-ceiling --profile-steps 0 > /dev/null 2>&1
python - <<'PY'
import glob, sqlite3
db = sorted(glob.glob("gpurun_out/prtl/kt/**/*.db", recursive=True))[0]
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("PRAGMA table_info(kernels)")]
recs = [dict(zip(cols, r)) for r in con.execute("SELECT * FROM kernels ORDER BY start")]
idx = [i for i, r in enumerate(recs) if "primal_kernel" in r["name"]]
a = idx[-10]; b = idx[-9]
t0 = recs[a]["start"]
for r in recs[a:b + 1]:
    print(f"{(r['start']-t0)/1e3:8.2f} -> {(r['end']-t0)/1e3:8.2f} us  {r['name'].replace('(anonymous namespace)::','')[:60]}  queue {r.get('queue_id', r.get('queue', '?'))}")
PY
