# dev: the sweep's knobs on the column-skewed 10M matrix (its A'y' product), one line per setting
mkdir -p gpurun_out/r5g
export SHAPE_CACHE_DIR=/tmp/shape_cache PDHG_DEV=1
O=gpurun_out/r5g/colskew_sweep.txt; : > $O
run() { echo "== $1" >> $O; timeout 300 python tools/shape_table.py --no-vendor --only "column-skewed" --env "$1" 2>&1 | grep -E "^column|steps per" | cut -c1-260 >> $O; }
run "PDHG_VERBOSE=1"
for c in 1.0 1.3 1.6 3.0; do run "PDHG_TW_NNZ_CAP=$c PDHG_VERBOSE=1"; done
for t in 256 512 1024; do run "PDHG_LONG_THR=$t PDHG_VERBOSE=1"; done
for m in 0 2; do run "PDHG_TW_MODE=$m"; done
run "PDHG_SPMV=stream"
cat $O
