# round 6 probe: the boundary between two trial graphs queued back to back (would a speculatively queued next trial start sooner?)
# (build the probe first: bash tools/variants.sh twice "-DPDHG_PROBE_GRAPH_TWICE")
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/prtl; mkdir -p gpurun_out/prtl
PDHG_HIP_LIB=$PWD/firstorderlp.jl_amd/csrc/variants/libpdhg_twice.so rocprofv3 --kernel-trace -d gpurun_out/prtl/kt -- python bench.py --workload pagerank --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs --no-self-profile --no-ceiling --no-vendor --no-details --profile-steps 0 > /dev/null 2>&1
python - <<'PY'
import glob, sqlite3, statistics
db = sorted(glob.glob("gpurun_out/prtl/kt/**/*.db", recursive=True))[0]
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("PRAGMA table_info(kernels)")]
recs = [dict(zip(cols, r)) for r in con.execute("SELECT * FROM kernels ORDER BY start")]
gaps = []
for a, b in zip(recs[:-1], recs[1:]):
    if "final_reduce_host" in a["name"] and "primal_kernel" in b["name"]:
        gaps.append((b["start"] - a["end"]) / 1e3)
gaps = gaps[20:]
print("final_reduce -> next primal gaps (us), alternating queued / after the host:", [round(g, 1) for g in gaps[:12]])
print("even", statistics.mean(gaps[0::2]), "odd", statistics.mean(gaps[1::2]))
PY
