# round 6: the persistent kernels' row trips with their epilogue operands requested beside the row pointers (-DPDHG_COH_EARLY_OPS)
export PDHG_DEV=1
V=$PWD/firstorderlp.jl_amd/csrc/variants
for rep in 1 2 3; do for lib in "" $V/libpdhg_earlyops.so; do
  PDHG_HIP_LIB=$lib python bench.py --workload l1svm --steps 4000 --warmup 300 --no-cpu-baseline --no-self-profile --no-vendor --no-details 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('l1svm lib=${lib##*/}', d['value'], d['ms_per_step'])"
done; done
for lib in "" $V/libpdhg_earlyops.so; do
  PDHG_HIP_LIB=$lib python bench.py --workload random --rows 100000 --cols 100000 --steps 3000 --warmup 300 --no-cpu-baseline --no-self-profile --no-vendor --no-details --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('random 100K lib=${lib##*/}', d['value'], d['ms_per_step'])"
done
