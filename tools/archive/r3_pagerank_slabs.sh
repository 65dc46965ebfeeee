#!/bin/bash
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
# PageRank-1M: where the boundary between the two column slabs sits (PDHG_SLAB_SPLIT = first slab's share of the columns)
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --workload pagerank --steps 2000 --warmup 300 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['layout']['A_slabs'], d['layout']['At_slabs'])"; }
run "equal halves"
for f in 0.15 0.25 0.35 0.42 0.6; do PDHG_SLAB_SPLIT=$f run "split=$f"; done
