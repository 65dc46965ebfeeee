#!/bin/bash
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
# optimize() end to end with solve_qp.jl's defaults, stage by stage (tools/solve_demo.py --breakdown): round 5, the three
# trust-region searches of a restart check in ONE persistent launch (pdhg_trust_region_bounds) against one launch each; the
# check's small reductions riding with pdhg_eval_point (PDHG_EVAL_PREFETCH) against their own launches
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5solve
O=gpurun_out/r5solve/r05_solve_demo.txt
{
echo "# optimize() end to end, solve_qp.jl defaults, 1e-4 (tools/archive/r5_solve_demo.sh; one fresh process per solve: each total includes ~0.15-0.25 s of first-launch costs)"
echo "# stages: wall clock around the host's calls (tools/solve_demo.py --breakdown)"
for args in "--workload random --n 1000000" "--workload pagerank --n 1000000" "--workload l1svm --iteration_limit 40000"; do
  echo "## $args"
  timeout 900 python tools/solve_demo.py $args --verbosity 0 --breakdown 2>/dev/null | tail -9
  echo "## $args   PDHG_EVAL_PREFETCH=0 (the restart distances and the sum of squares launched and waited for by their own calls)"
  PDHG_EVAL_PREFETCH=0 timeout 900 python tools/solve_demo.py $args --verbosity 0 --breakdown 2>/dev/null | tail -9
  echo "## $args   PDHG_EVAL_PREFETCH=0 PDHG_TR_BATCH=0 (round 4: one persistent launch per search)"
  PDHG_EVAL_PREFETCH=0 PDHG_TR_BATCH=0 timeout 900 python tools/solve_demo.py $args --verbosity 0 --breakdown 2>/dev/null | tail -9
done
echo "## l1svm --no-record (only the terminating check's stats are kept)"
timeout 900 python tools/solve_demo.py --workload l1svm --iteration_limit 40000 --verbosity 0 --breakdown --no-record 2>/dev/null | tail -9
echo "## l1svm --no-record PDHG_TR_BATCH=0"
PDHG_TR_BATCH=0 timeout 900 python tools/solve_demo.py --workload l1svm --iteration_limit 40000 --verbosity 0 --breakdown --no-record 2>/dev/null | tail -9
} > $O
cat $O
