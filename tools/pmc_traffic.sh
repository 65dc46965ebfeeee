#!/bin/bash
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
# HBM/fabric bytes per fused product from hardware counters, the way MI355X_MICROARCH.md
# prescribes: separate `rocprofv3 --pmc` passes (kernel-trace only) of the SAME bench.py
# command with the separate-launch path (PDHG_GRAPH=0: the kernels are the ones the one-launch
# paths run, launched one by one so that the counters attribute to them), read requests by
# size (32 B / 128 B / the rest at 64 B) plus WRITE_SIZE (KiB), summed over the kernels of each
# product (pdhg_kernel_name's " + " list).  Run on the GPU box:
#     tools/pmc_traffic.sh "<round label>" [workload ...]      (default: random pagerank l1svm)
# Writes gpurun_out/pmc_traffic/pmc_traffic.json: copy it to profiles/pmc_traffic.json
# (bench.py reports it as roofline.traffic when workload and kernel group match).
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_traffic
LABEL="${1:-round 3}"; shift
WLS="${@:-random pagerank l1svm}"
rm -rf $O; mkdir -p $O
for WL in $WLS; do
  B="python $R/bench.py --workload $WL --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --profile-steps 0 --no-self-profile --full-line --no-details"
  PDHG_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum -d $O/$WL/p1 -- $B > $O/$WL.p1.log 2>&1 || echo "pass 1 failed"
  PDHG_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum -d $O/$WL/p2 -- $B > $O/$WL.p2.log 2>&1 || echo "pass 2 failed"
  PDHG_GRAPH=0 timeout 600 $B > $O/$WL.line.json 2> $O/$WL.line.err
done
cd $R
python - "$O" "$LABEL" $WLS <<'PY'
import glob, json, os, re, sqlite3, sys
root, label, wls = sys.argv[1], sys.argv[2], sys.argv[3:]
out = {"_comment": "HBM/fabric bytes per fused product from rocprofv3 PMC passes (tools/pmc_traffic.sh: separate --pmc runs of "
                   "`PDHG_GRAPH=0 bench.py --workload W --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --profile-steps 0`; "
                   "reads = 32*RDREQ_32B + 128*RDREQ_128B + 64*(RDREQ - RDREQ_32B - RDREQ_128B), writes = 1024*WRITE_SIZE, per "
                   "launch, summed over the kernels of the product with their launches per product; the formula reproduces the "
                   "elementwise kernels' algorithmic bytes). bench.py reports these as roofline.traffic when workload and kernel "
                   "group match.",
       "_round": label, "_detail": {}}
def short(full):
    m = re.search(r"([A-Za-z_0-9]+(?:<[^>]*>)?)\(", full)
    return m.group(1) if m else full
for wl in wls:
    vals, calls = {}, {}
    for db in glob.glob(os.path.join(root, wl, "**", "*.db"), recursive=True):
        con = sqlite3.connect(db)
        for k, c, n, v in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                      "group by kernel_name, counter_name"):
            vals.setdefault(short(k), {})[c] = v
            calls[short(k)] = max(calls.get(short(k), 0), n)
    try:
        line = json.loads(open(os.path.join(root, wl + ".line.json")).read().strip().splitlines()[-1])
    except Exception as exc:
        print("no bench line for", wl, exc); continue
    m, n, nnz = line["config"]["m"], line["config"]["n"], line["config"]["nnz"]
    per_kernel = {}
    for k, c in vals.items():
        if "TCC_EA0_RDREQ_sum" not in c:
            continue
        r32, r128, r = c.get("TCC_EA0_RDREQ_32B_sum", 0.0), c.get("TCC_EA0_RDREQ_128B_sum", 0.0), c["TCC_EA0_RDREQ_sum"]
        reads = 32 * r32 + 128 * r128 + 64 * (r - r32 - r128)
        writes = 1024.0 * c.get("WRITE_SIZE", 0.0)
        per_kernel[k] = {"read_bytes": int(reads), "write_bytes": int(writes), "read_requests": int(r),
                         "l2_hits": int(c.get("TCC_HIT_sum", 0)), "l2_misses": int(c.get("TCC_MISS_sum", 0)), "launches": calls[k]}
    out["_detail"][wl] = per_kernel
    for labelk in line.get("layout_products", []):
        members = [x.strip() for x in labelk.split("+")]
        if not all(x in per_kernel for x in members):
            print("missing counters for", labelk); continue
        base = min(per_kernel[x]["launches"] for x in members if "long" not in x)
        total = sum((per_kernel[x]["read_bytes"] + per_kernel[x]["write_bytes"]) * per_kernel[x]["launches"] / base for x in members)
        out[f"{labelk}@m={m},n={n},nnz={nnz}"] = int(total)
json.dump(out, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))
PY
