set -x
bash tools/r6_bench_check.sh --steps 20 --warmup 5 2>&1 | tail -5
python -m pytest tests/test_gpu_sj.py tests/test_gpu_rescale.py -x -q 2>&1 | tail -5
