# dev: spmv_sj_kernel by epilogue mode on one shape, kernel times from rocprofv3 (every step under its own timeout)
R=${GRAFT_REPO_ROOT:-$PWD}
export SHAPE_CACHE_DIR=/tmp/shape_cache PDHG_DEV=1 PDHG_GRAPH=0 PDHG_COOP=0
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sjm
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/sjm -- python $R/tools/archive/r5_sj_modes.py "${1:-blockdiag}" > /tmp/sjm.log 2>&1 < /dev/null
timeout 60 python $R/tools/rocprof_summary.py /tmp/sjm < /dev/null > /tmp/sjm.json 2>/dev/null
timeout 30 python - < /dev/null <<'PY'
import json
d = json.load(open("/tmp/sjm.json"))
for k in d.get("kernels", []):
    n = k["name"]
    if "spmv_sj" in n or "spmv_stream" in n or "spmv_long" in n:
        i = n.find("spmv_")
        print(n[i:n.find("(", i)], k["calls"], round(k["avg_us"], 1))
PY
