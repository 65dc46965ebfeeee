export PDHG_DEV=1
mkdir -p gpurun_out/r6prof
timeout 600 python tools/archive/r6_qp_sj_time.py 2>&1 | tail -12
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/r6_bench_check.sh --steps 20 --warmup 5 | tail -3 | cut -c1-300
cp gpurun_out/r6/bench_stdout.txt gpurun_out/r6prof/r06_bench_default_line.json
cp gpurun_out/r6/bench_details.json gpurun_out/r6prof/r06_bench_default.json
( time timeout 900 python -m pytest tests/test_gpu_sj.py tests/test_gpu_fake_rccl.py -x -q --durations=5 2>&1 | tail -12 ) 2>&1
