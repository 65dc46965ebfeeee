export PDHG_DEV=1 SHAPE_CACHE_DIR=/tmp/shapes
mkdir -p gpurun_out/r6 /tmp/shapes
T=gpurun_out/r6/sj_wide3.txt
: > $T
runhere() { echo "## $1 | env: $2" >> $T; env $2 timeout 900 python tools/shape_table.py --only "$1" $3 2>&1 | grep -v "^#" >> $T; }
for shape in "banded 10M +-50000" "blockdiag 10M"; do
  runhere "$shape" "PDHG_SJ_WIDE=0" "--no-vendor"
  runhere "$shape" "PDHG_SJ_WIDE=1" "--no-vendor"
done
cat $T | cut -c1-200
timeout 2400 python -m pytest tests/test_gpu_fake_rccl.py -x -q -k "all_gather or overlapped" 2>&1 | tail -15
