mkdir -p gpurun_out/r5c
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
(tools/slot_probe 100 100000 8; tools/slot_probe 100 100000 4) > gpurun_out/r5c/slot_probe4.txt 2>&1
grep "^  1 \|^  12\|^entries" gpurun_out/r5c/slot_probe4.txt
export SHAPE_CACHE_DIR=/tmp/shape_cache
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r5c/vendor_kt -- python $R/tools/vendor_spmv.py --shape banded50k > $R/gpurun_out/r5c/vendor_kt.log 2>&1
cd $R
python - <<'PY'
import glob, sqlite3
for db in glob.glob('gpurun_out/r5c/vendor_kt/**/*.db', recursive=True):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if 'kernel' in t.lower()]
    print(kd[:10])
    try:
        cols = [r[1] for r in con.execute("PRAGMA table_info(kernels)")]
        print(cols)
        q = "select name, count(*), avg(end-start), min(grid_x), min(workgroup_x), min(lds_size), min(vgpr_count) from kernels group by name order by 3 desc"
        for r in con.execute(q).fetchall()[:25]:
            print(r)
    except Exception as e:
        print("err", e)
PY
tail -5 gpurun_out/r5c/vendor_kt.log
