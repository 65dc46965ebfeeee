export PDHG_DEV=1
mkdir -p gpurun_out/r6
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=12 2>&1 | tail -30 ) 2>&1 | tee gpurun_out/r6/full_gpu_suite_final.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
