# round 6: the wide sliced jagged form + hub rows -- tests, then PageRank-1M / banded / block-diagonal 10M under the variants
export PDHG_DEV=1 SHAPE_CACHE_DIR=/tmp/shapes
mkdir -p gpurun_out/r6 /tmp/shapes
timeout 1500 python -m pytest tests/test_gpu_sj.py -x -q 2>&1 | tail -8
T=gpurun_out/r6/sj_wide_table.txt
: > $T
run() { echo "## $1 | env: $2" >> $T; timeout 600 python tools/shape_table.py --only "$1" $3 --env "$2" 2>&1 | grep -v "^#" >> $T; }
run "pagerank 1M" "" ""
run "pagerank 1M" "PDHG_SJ=0" "--no-vendor"
run "pagerank 1M" "PDHG_SLABS=0" "--no-vendor"
run "pagerank 1M" "PDHG_SLABS=0 PDHG_SJ=0" "--no-vendor"
run "pagerank 1M" "PDHG_SJ_MAXLEN=64" "--no-vendor"
run "pagerank 1M" "PDHG_SJ_MAXLEN=256" "--no-vendor"
run "pagerank 1M" "PDHG_SJ_MAXLEN=64 PDHG_SLABS=0" "--no-vendor"
run "pagerank 1M" "PDHG_SJ=1 PDHG_SJ_WIDE=0" "--no-vendor"
run "banded 10M +-50000" "" "--no-vendor"
run "banded 10M +-50000" "PDHG_SJ_WIDE=0" "--no-vendor"
run "banded 10M +-50000" "PDHG_SJ_WIDE=1" "--no-vendor"
run "blockdiag 10M" "" "--no-vendor"
run "blockdiag 10M" "PDHG_SJ_WIDE=0" "--no-vendor"
run "blockdiag 10M" "PDHG_SJ_WIDE=1" "--no-vendor"
run "4M-pagerank" "" ""
run "4M-pagerank" "PDHG_SJ=0" "--no-vendor"
cat $T
