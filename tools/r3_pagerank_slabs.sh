#!/bin/bash
# PageRank-1M: slab size of the stream layout's column-slab passes
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --workload pagerank --steps 2000 --warmup 300 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['layout']['A_slabs'], d['layout']['At_slabs'])"; }
for mb in 4 2.7 3.3 5 8.1; do PDHG_SLAB_MB=$mb run "slab_mb=$mb"; done
PDHG_SLABS=0 run "no slabs"
timeout 600 python -m pytest tests/test_gpu_dist_group.py -x -q -k "batched_take_steps" 2>&1 | tail -2
