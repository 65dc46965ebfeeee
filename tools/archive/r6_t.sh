# round 6: rows per row block capped at 512 / 768 instead of 1 024 (blocks of very short rows take fewer row trips) on the L1-SVM LP
export PDHG_DEV=1
V=$PWD/firstorderlp.jl_amd/csrc/variants
for rep in 1 2; do for lib in "" $V/libpdhg_rows768.so $V/libpdhg_rows512.so; do
  PDHG_HIP_LIB=$lib python bench.py --workload l1svm --steps 4000 --warmup 300 --no-cpu-baseline --no-self-profile --no-vendor --no-details 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('l1svm lib=${lib##*/}', d['value'], d['ms_per_step'], d['launch_path'][:40])"
done; done
for lib in "" $V/libpdhg_rows512.so; do
  PDHG_HIP_LIB=$lib python bench.py --workload pagerank --steps 3000 --warmup 300 --no-cpu-baseline --no-self-profile --no-vendor --no-details 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('pagerank lib=${lib##*/}', d['value'], d['ms_per_step'])"
done
