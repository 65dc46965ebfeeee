export PDHG_DEV=1 SHAPE_CACHE_DIR=/tmp/shapes
mkdir -p gpurun_out/r6 /tmp/shapes
timeout 1500 python -m pytest tests/test_gpu_sj.py -x -q 2>&1 | tail -8
T=gpurun_out/r6/sj_wide2.txt
: > $T
runhere() { echo "## $1 | env: $2" >> $T; env $2 timeout 900 python tools/shape_table.py --only "$1" $3 2>&1 | grep -v "^#" >> $T; }
for shape in "banded 10M +-50000" "blockdiag 10M"; do
  runhere "$shape" "PDHG_SJ_WIDE=0" "--no-vendor"
  runhere "$shape" "PDHG_SJ_WIDE=1" "--no-vendor"
  runhere "$shape" "" "--no-vendor"
done
runhere "pagerank 1M" "" "--no-vendor"
runhere "1M-banded +-5000, lognormal rows" "" ""
runhere "1M-banded +-5000, lognormal rows" "PDHG_SJ=0" "--no-vendor"
runhere "10M-banded +-50000, lognormal rows" "" ""
runhere "10M-banded +-50000, lognormal rows" "PDHG_SJ=0" "--no-vendor"
runhere "10M-banded +-50000, lognormal rows" "PDHG_SJ=1 PDHG_SJ_WIDE=1" "--no-vendor"
cat $T
