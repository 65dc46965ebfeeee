# round 6: PageRank-1M, column slabs re-measured on the final stream kernel (slabs off / slab sizes)
export PDHG_DEV=1
run() { python bench.py --workload pagerank --steps 3000 --warmup 300 --no-cpu-baseline --no-self-profile --no-vendor --no-details 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run default
PDHG_SLABS=0 run slabs_off
PDHG_SLAB_MB=2 run slab_2MB
PDHG_SLAB_MB=3 run slab_3MB
PDHG_SLAB_MB=6 run slab_6MB
done
