#!/bin/bash
export PDHG_DEV=1
# the sweep with the entry loads of B steps requested at once (-DPDHG_TW_BURST=B variants of the library) against the shipped order
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5g
export SHAPE_CACHE_DIR=/tmp/shape_cache
V=firstorderlp.jl_amd/csrc/variants
PDHG_HIP_LIB=$PWD/$V/libpdhg_burst4.so timeout 900 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_full_size.py -x -q 2>&1 | tail -4
for b in 0 2 3 4; do
  echo "== burst $b"
  PDHG_HIP_LIB=$PWD/$V/libpdhg_burst$b.so python tools/shape_table.py --no-vendor --only "uniform 1M,uniform 4M,config S,30 per row,column-skewed,clustered" | grep -v "^#"
done
