# rocprofv3 --kernel-trace --stats of the shape-table runs of the review's non-uniform shapes (every step under a timeout,
# stdin closed): the kernel times behind profiles/r05_shape_table.txt's HIP-event brackets
R=${GRAFT_REPO_ROOT:-$PWD}
export SHAPE_CACHE_DIR=/tmp/shape_cache PDHG_DEV=1
mkdir -p $R/gpurun_out/r5n
cd /tmp && export TMPDIR=/tmp
for S in "clustered 10M" "column-skewed 10M" "banded 10M +-3000000" "10M-lognormal"; do
  T=$(echo "$S" | tr -c 'A-Za-z0-9' '_')
  rm -rf /tmp/nu_$T
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/nu_$T -- python $R/tools/shape_table.py --no-vendor --only "$S" > $R/gpurun_out/r5n/$T.txt 2>/dev/null < /dev/null
  timeout 60 python $R/tools/rocprof_summary.py /tmp/nu_$T < /dev/null > $R/gpurun_out/r5n/$T.json 2>/dev/null
  grep -v "^#" $R/gpurun_out/r5n/$T.txt | cut -c1-170
done
timeout 60 python - < /dev/null <<PY
import json, glob, os
out = {"what": "rocprofv3 --kernel-trace --stats around tools/shape_table.py --no-vendor --only <shape> (35 take_steps each: 5 untimed + 30 bracketed), final build; spmv_* kernels only; the shape table's HIP-event brackets include ~10 us of bracket overhead per product",
       "shapes": {}}
for f in sorted(glob.glob("$R/gpurun_out/r5n/*.json")):
    try:
        d = json.load(open(f))
    except Exception:
        continue
    rows = [{"kernel": k["name"][k["name"].find("spmv_"):k["name"].find("(", k["name"].find("spmv_"))], "calls": k["calls"], "avg_us": round(k["avg_us"], 1)}
            for k in d.get("kernels", []) if "spmv_" in k["name"]]
    out["shapes"][os.path.basename(f)[:-5]] = rows
    print(os.path.basename(f)[:-5], rows)
json.dump(out, open("$R/gpurun_out/r5n/r05_nonuniform_rocprof_summary.json", "w"), indent=1)
PY
