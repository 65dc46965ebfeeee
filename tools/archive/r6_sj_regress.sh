# round 6: where did the narrow sliced jagged kernel lose 14 %?  r5's tree against the current one and compile-time variants
export PDHG_DEV=1 SHAPE_CACHE_DIR=/tmp/shapes
mkdir -p gpurun_out/r6 /tmp/shapes
T=gpurun_out/r6/sj_waitcnt.txt
: > $T
runhere() { echo "## $1 | current | env: $2" >> $T; env $2 timeout 900 python tools/shape_table.py --only "$1" --no-vendor 2>&1 | grep -v "^#" >> $T; }
V=$PWD/firstorderlp.jl_amd/csrc/variants
for shape in "banded 10M +-50000" "blockdiag 10M"; do
  runhere "$shape" "PDHG_SJ_WIDE=0"
  test -d .r5tree && { echo "## $shape | r5 tree" >> $T; (cd .r5tree && timeout 900 python tools/shape_table.py --only "$shape" --no-vendor 2>&1 | grep -v "^#") >> $T; }
  runhere "$shape" "PDHG_SJ_WIDE=0 PDHG_HIP_LIB=$V/libpdhg_nowait.so"
  runhere "$shape" "PDHG_SJ_WIDE=1"
  runhere "$shape" ""
done
runhere "pagerank 1M" ""
runhere "pagerank 1M" "PDHG_SJ_MAXLEN=64 PDHG_SLABS=0"
runhere "pagerank 1M" "PDHG_SJ_MAXLEN=32 PDHG_SLABS=0"
runhere "pagerank 1M" "PDHG_SJ=0"
cat $T
