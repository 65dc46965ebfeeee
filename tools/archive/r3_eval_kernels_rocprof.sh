#!/bin/bash
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for arg in l1svm 1000000; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/g28_$arg -o eval -- python tools/eval_cost.py $arg > /dev/null 2>&1
f=$(find gpurun_out/g28_$arg -name "*kernel_stats.csv" | head -1)
echo "== $arg $f"; head -16 "$f" | cut -d, -f1-8
done
