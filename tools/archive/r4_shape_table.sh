#!/bin/bash
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
# the two fused products on a range of matrix shapes with the shipped build (tools/tune_tiled.py: HIP-event brackets of
# pdhg_trial_step's products from a fixed iterate); round 2's table (profiles/r02_final_shape_table.txt) re-measured
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r4shape
O=gpurun_out/r4shape/r04_shape_table.txt
: > $O
run() { echo "== $1" >> $O; shift; timeout 600 python tools/tune_tiled.py --reps 1 "$@" 2>/dev/null | grep "^rep" >> $O; }
run "uniform 1M x 1M, 10 per row" --m 1000000 --n 1000000
run "uniform 4M x 4M, 10 per row" --m 4000000 --n 4000000
run "uniform 10M x 10M, 10 per row (config S)"
run "uniform 10M x 10M, 30 per row" --k 30
run "tall 10M x 1M, 10 per row" --m 10000000 --n 1000000
run "column-skewed 10M" --colskew
run "banded 10M +-50000" --banded 50000
run "banded 10M +-3000000" --banded 3000000
run "blockdiag 10M" --shape blockdiag
run "clustered 10M" --shape clustered
run "twodensity 10M" --shape twodensity
run "arrowhead 10M" --shape arrowhead
run "pagerank 1M" --pagerank 1000000
cat $O
