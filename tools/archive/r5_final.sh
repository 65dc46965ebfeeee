#!/bin/bash
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
# round 5's final measurement set: bench lines + rocprofv3 summaries + PMC traffic (tools/archive/r5_profile.sh), the shape table with
# the vendor's kernel beside every product, counter passes of the sliced jagged kernel on the banded matrix
cd "$GRAFT_REPO_ROOT"
bash tools/archive/r5_profile.sh > gpurun_out/r5prof.log 2>&1
export SHAPE_CACHE_DIR=/tmp/shape_cache
mkdir -p gpurun_out/r5f
python tools/shape_table.py > gpurun_out/r5f/r05_shape_table.txt 2> gpurun_out/r5f/shape.err
bash tools/pmc_stream.sh "banded 10M +-50000" banded50k_sj spmv_sj_kernel > gpurun_out/r5f/pmc_sj.log 2>&1
tail -3 gpurun_out/r5prof.log; cut -c1-200 gpurun_out/r5f/r05_shape_table.txt
