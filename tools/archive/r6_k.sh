export PDHG_DEV=1
V=$PWD/firstorderlp.jl_amd/csrc/variants
PDHG_HIP_LIB=$V/libpdhg_trbtrace.so timeout 600 python tools/solve_demo.py --workload l1svm --iteration_limit 40000 --verbosity 0 --breakdown 2>&1 | tail -10
timeout 600 python tools/solve_demo.py --workload l1svm --iteration_limit 40000 --verbosity 0 --breakdown 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_trust_region.py tests/test_gpu_device_eval.py -x -q 2>&1 | tail -4
