export PDHG_DEV=1 SHAPE_CACHE_DIR=/tmp/shapes
mkdir -p gpurun_out/r6 /tmp/shapes
T=gpurun_out/r6/stream_waitcnt2.txt
: > $T
V=$PWD/firstorderlp.jl_amd/csrc/variants
for lib in "" $V/libpdhg_signed.so $V/libpdhg_nogwait.so $V/libpdhg_r5like.so; do
  for wl in pagerank l1svm; do
    echo "## bench --workload $wl lib=${lib##*/}" >> $T
    for rep in 1 2; do
    PDHG_HIP_LIB=$lib python bench.py --workload $wl --steps 4000 --warmup 300 --no-cpu-baseline --no-self-profile --no-vendor --no-details 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" >> $T
    done
  done
  echo "## shapes lib=${lib##*/}" >> $T
  PDHG_HIP_LIB=$lib python tools/shape_table.py --only "pagerank 1M,l1svm" --no-vendor 2>&1 | grep -v "^#" | cut -c1-200 >> $T
done
cat $T
