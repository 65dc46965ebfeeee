#!/bin/bash
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
# optimize() end to end with solve_qp.jl's defaults on the four benchmark LPs, stage by stage (tools/solve_demo.py --breakdown)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4solve
O=gpurun_out/r4solve/r04_solve_demo.txt
{
echo "# optimize() end to end, solve_qp.jl defaults, 1e-4 (tools/r4_solve_demo.sh; one fresh process per solve: each total includes ~0.15-0.25 s of first-launch costs)"
echo "# stages: wall clock around the host's calls (tools/solve_demo.py --breakdown); round 4: trust-region searches as one persistent launch (n + m <= 1M), eval_point in one round trip"
for args in "--workload random --n 10000000 --iteration_limit 60000" "--workload random --n 1000000" "--workload pagerank --n 1000000" "--workload l1svm --iteration_limit 40000"; do
  timeout 900 python tools/solve_demo.py $args --verbosity 0 --breakdown 2>/dev/null | tail -9
done
echo "# PDHG_TR_COOP=0 (round 3's pass-by-pass searches), L1-SVM"
PDHG_TR_COOP=0 timeout 900 python tools/solve_demo.py --workload l1svm --iteration_limit 40000 --verbosity 0 --breakdown 2>/dev/null | tail -9
} > $O
cat $O
