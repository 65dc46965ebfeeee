#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for n in 1000000; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/g32_$n -o b -- python bench.py --m $n --n $n --steps 1500 --warmup 100 --no-cpu-baseline --no-other-configs --profile-steps 0 > gpurun_out/g32_$n.json 2>/dev/null
python - <<PY
import csv, json
d=json.loads(open("gpurun_out/g32_$n.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["layout"])
tot=0
for r in csv.DictReader(open("gpurun_out/g32_$n/b_kernel_stats.csv")):
    if int(r["Calls"])>=1500: print(r["Name"][:70], r["Calls"], r["AverageNs"]); tot+=float(r["AverageNs"])*int(r["Calls"])/1600
print("sum per iteration us", tot/1e3)
PY
done
