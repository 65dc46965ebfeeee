export PDHG_DEV=1 SHAPE_CACHE_DIR=/tmp/shapes
mkdir -p gpurun_out/r6 /tmp/shapes
T=gpurun_out/r6/sj_uniform_1m.txt
: > $T
for env in "" "PDHG_SPMV=stream" "PDHG_SPMV=stream PDHG_SJ=1" "PDHG_SPMV=stream PDHG_SJ=1 PDHG_SLAB_MB=2" "PDHG_SPMV=stream PDHG_SJ=1 PDHG_SJ_WIDE=1" "PDHG_SPMV=stream PDHG_SJ=1 PDHG_SLABS=0"; do
  echo "## env: $env" >> $T
  env $env python tools/shape_table.py --only "uniform 1M x 1M" --no-vendor 2>&1 | grep -v "^#" | cut -c1-220 >> $T
done
cat $T
