# Kernel timeline of PageRank-1M trials on the HIP-graph path (rocprofv3 --kernel-trace): one trial printed, and over all
# traced trials the WALL time of each product on the critical path (first pass start -> last pass end; the long-row pair
# runs beside the passes on a second hardware queue) (their durations under the trace are stretched by that concurrency: the stats run has them at 7.4 + 5.0 / 5.3 + 4.9 us alone).
# Usage (GPU box): bash tools/pagerank_trial_timeline.sh > gpurun_out/pagerank_trial_timeline.txt
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/prtl; mkdir -p gpurun_out/prtl
rocprofv3 --kernel-trace -d gpurun_out/prtl/kt -- python bench.py --workload pagerank --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --no-self-profile --no-ceiling --no-vendor --no-details --profile-steps 0 > /dev/null 2>&1
python - <<'PY'
import glob, sqlite3, statistics
db = sorted(glob.glob("gpurun_out/prtl/kt/**/*.db", recursive=True))[0]
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("PRAGMA table_info(kernels)")]
recs = [dict(zip(cols, r)) for r in con.execute("SELECT * FROM kernels ORDER BY start")]
for r in recs:
    r["name"] = r["name"].replace("(anonymous namespace)::", "")
idx = [i for i, r in enumerate(recs) if "primal_kernel" in r["name"]]
a, b = idx[-10], idx[-9]
t0 = recs[a]["start"]
print("# one trial (us from its first kernel's start)")
for r in recs[a:b + 1]:
    print(f"{(r['start']-t0)/1e3:8.2f} -> {(r['end']-t0)/1e3:8.2f} us  {r['name'][:64]}  queue {r.get('queue_id', r.get('queue', '?'))}")
wall = {0: [], 1: []}; ksum = {0: [], 1: []}; trial = []; gaps = []
for a, b in zip(idx[20:-1], idx[21:]):
    ks = recs[a:b]
    if len(ks) != 10:
        continue      # a trial with another shape (a check between two batches)
    for tag in (0, 1):
        passes = [k for k in ks if "spmv_stream_kernel" in k["name"] and k["name"].rstrip().split("(")[0].endswith(f"{tag}>")]
        longs = [k for k in ks if "spmv_long" in k["name"] and (("partial_kernel<%d>" % tag) in k["name"] or ("final_kernel<%d>" % (tag + 1)) in k["name"])]
        if len(passes) != 2:
            break
        wall[tag].append((max(k["end"] for k in passes) - min(k["start"] for k in passes)) / 1e3)
        ksum[tag].append(sum(k["end"] - k["start"] for k in passes + longs) / 1e3)
    else:
        trial.append((recs[b]["start"] - ks[0]["start"]) / 1e3)
        gaps.append(trial[-1] - sum(k["end"] - k["start"] for k in ks if "spmv_long" not in k["name"]) / 1e3)
print(f"# {len(trial)} trials of 10 kernels")
for tag, nm in ((0, "A xbar"), (1, "A'y'")):
    print(f"{nm:7s}: wall on the critical path {statistics.mean(wall[tag]):6.2f} us (median {statistics.median(wall[tag]):6.2f})")
print(f"trial  : {statistics.mean(trial):6.2f} us start to start under the profiler, of which {statistics.mean(gaps):5.2f} us between kernels")
PY
