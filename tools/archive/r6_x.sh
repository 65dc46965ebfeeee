# round 6: the next trial queued behind the current one (PDHG_SPEC_QUEUE) on the graph path: PageRank-1M A/B
# (these two scripts measured a build that is no longer in the tree: NOTEBOOK section 10.5)
export PDHG_DEV=1
run() { python bench.py --workload pagerank --steps 3000 --warmup 300 --no-cpu-baseline --no-self-profile --no-vendor --no-details 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('pagerank $1', d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
PDHG_SPEC_QUEUE=0 run off
PDHG_SPEC_QUEUE=1 run on
done
