export PDHG_DEV=1
V=$PWD/firstorderlp.jl_amd/csrc/variants
PDHG_HIP_LIB=$V/libpdhg_trbtrace.so timeout 600 python tools/solve_demo.py --workload l1svm --iteration_limit 40000 --verbosity 0 --breakdown 2>&1 | tail -14
PDHG_HIP_LIB=$V/libpdhg_trbtrace.so timeout 600 python tools/solve_demo.py --workload random --n 1000000 --verbosity 0 --breakdown 2>&1 | tail -12
