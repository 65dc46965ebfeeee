#!/bin/bash
# round 6, last validation on the final build: the full GPU suite, smoke(), and the driver's bench command checked the way
# the driver reads it (tools/r6_bench_check.sh)
export PDHG_DEV=1
mkdir -p gpurun_out/r6 gpurun_out/r6prof
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -24 ) 2>&1 | tee gpurun_out/r6/full_gpu_suite_final.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/r6_bench_check.sh --steps 20 --warmup 5 | tail -3 | cut -c1-400
cp gpurun_out/r6/bench_stdout.txt gpurun_out/r6prof/r06_bench_default_line.json
cp gpurun_out/r6/bench_details.json gpurun_out/r6prof/r06_bench_default.json
