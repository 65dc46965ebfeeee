#!/bin/bash
# HBM/fabric bytes per launch of the hot kernels from hardware counters, the way
# MI355X_MICROARCH.md prescribes: separate `rocprofv3 --pmc` passes (kernel-trace only) of
# the SAME bench.py command, read requests by size (32 B / 128 B / the rest at 64 B) plus
# WRITE_SIZE (KiB).  Run on the GPU box:  tools/pmc_traffic.sh [round-label]
# Writes gpurun_out/pmc_traffic/{p1,p2}/ and gpurun_out/pmc_traffic/pmc_traffic.json
# (copy that file to profiles/pmc_traffic.json: bench.py reports it as roofline.traffic).
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_traffic
rm -rf $O; mkdir -p $O
B="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --profile-steps 0"
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum -d $O/p1 -- $B > $O/p1.log 2>&1 || echo "pass 1 failed"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum -d $O/p2 -- $B > $O/p2.log 2>&1 || echo "pass 2 failed"
cd $R
python - "$O" "${1:-round 2}" <<'PY'
import glob, json, os, sqlite3, sys
root, label = sys.argv[1], sys.argv[2]
vals = {}
for db in glob.glob(os.path.join(root, "**", "*.db"), recursive=True):
    con = sqlite3.connect(db)
    for k, c, n, v in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                  "group by kernel_name, counter_name"):
        vals.setdefault(k, {})[c] = v
out = {"_comment": "HBM/fabric bytes per launch from rocprofv3 PMC passes (tools/pmc_traffic.sh: separate --pmc runs of "
                   "`bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --profile-steps 0`; reads = "
                   "32*RDREQ_32B + 128*RDREQ_128B + 64*(RDREQ - RDREQ_32B - RDREQ_128B), writes = 1024*WRITE_SIZE; "
                   "the formula reproduces the elementwise kernels' algorithmic bytes). bench.py reports these as "
                   "roofline.traffic when workload and kernel match.",
       "_round": label, "_detail": {}}
names = {"spmv_tiled_kernel<1": "spmv_tiled_kernel<MODE_DUAL>", "spmv_tiled_kernel<2": "spmv_tiled_kernel<MODE_ATY>",
         "primal_kernel": "primal_kernel", "accept_kernel": "accept_kernel"}
for k, c in vals.items():
    for needle, nice in names.items():
        if needle in k and "TCC_EA0_RDREQ_sum" in c:
            r32, r128, r = c.get("TCC_EA0_RDREQ_32B_sum", 0.0), c.get("TCC_EA0_RDREQ_128B_sum", 0.0), c["TCC_EA0_RDREQ_sum"]
            reads = 32 * r32 + 128 * r128 + 64 * (r - r32 - r128)
            writes = 1024.0 * c.get("WRITE_SIZE", 0.0)
            key = nice + "@m=10000000,n=10000000,nnz=100000000"
            if nice.startswith("spmv"):
                out[key] = int(reads + writes)
            out["_detail"][nice] = {"read_bytes": int(reads), "write_bytes": int(writes), "read_requests": int(r),
                                    "l2_hits": int(c.get("TCC_HIT_sum", 0)), "l2_misses": int(c.get("TCC_MISS_sum", 0))}
json.dump(out, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
