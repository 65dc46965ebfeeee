# round 6: kernel trace of PageRank-1M trials with the next trial queued behind the current one
# (these two scripts measured a build that is no longer in the tree: NOTEBOOK section 10.5)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/prtl; mkdir -p gpurun_out/prtl
rocprofv3 --kernel-trace -d gpurun_out/prtl/kt -- python bench.py --workload pagerank --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs --no-self-profile --no-ceiling --no-vendor --no-details --profile-steps 0 > /dev/null 2>&1
python - <<'PY'
import glob, sqlite3, statistics
db = sorted(glob.glob("gpurun_out/prtl/kt/**/*.db", recursive=True))[0]
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("PRAGMA table_info(kernels)")]
recs = [dict(zip(cols, r)) for r in con.execute("SELECT * FROM kernels ORDER BY start")]
for r in recs: r["name"] = r["name"].replace("(anonymous namespace)::", "")
idx = [i for i, r in enumerate(recs) if "primal_" in r["name"]]
a, b = idx[-10], idx[-9]
t0 = recs[a]["start"]
for r in recs[a:b + 1]:
    print(f"{(r['start']-t0)/1e3:8.2f} -> {(r['end']-t0)/1e3:8.2f} us  {r['name'][:70]}")
gaps = []; dur = []
for x, y in zip(recs[:-1], recs[1:]):
    if "final_reduce_host" in x["name"] and "primal_gated" in y["name"]:
        gaps.append((y["start"] - x["end"]) / 1e3); dur.append((y["end"] - y["start"]) / 1e3)
print("gaps", [round(g, 1) for g in gaps[40:80]])
print("durs", [round(g, 1) for g in dur[40:80]])
tt = [ (recs[j]["start"] - recs[i]["start"]) / 1e3 for i, j in zip(idx[40:80], idx[41:81])]
print("trial start to start", [round(g, 1) for g in tt])
print("final_reduce -> primal_gated gap", statistics.mean(gaps), "primal_gated duration", statistics.mean(dur), statistics.median(dur), len(gaps))
PY
