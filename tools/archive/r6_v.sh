# round 6: row blocks closed at fewer entries than BLOCK_NNZ (PDHG_BLOCK_CAP, dev): PageRank-1M, L1-SVM, random 100K
export PDHG_DEV=1
run() { python bench.py --workload $2 --steps $3 --warmup 300 --no-cpu-baseline --no-self-profile --no-vendor --no-details 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$2 $1', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for cap in 2048 1536 1024 768 512; do PDHG_BLOCK_CAP=$cap run cap$cap pagerank 3000; done
done
for cap in 2048 1024 512; do PDHG_BLOCK_CAP=$cap run cap$cap l1svm 4000; done
