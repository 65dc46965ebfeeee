# round 6: the shared wave reduction in the check kernels: tests of the check entry points, then what a check costs in a
# whole solve (L1-SVM, random 1M, PageRank-1M) against the one-tree-per-quantity build
export PDHG_DEV=1
V=$PWD/firstorderlp.jl_amd/csrc/variants
timeout 900 python -m pytest tests/test_gpu_trust_region.py tests/test_gpu_device_eval.py tests/test_gpu_end_to_end.py tests/test_gpu_kat.py tests/test_gpu_rescale.py -x -q 2>&1 | tail -3
for w in "--workload l1svm --iteration_limit 40000" "--workload random --n 1000000" "--workload pagerank --n 1000000"; do
  for lib in "" $V/libpdhg_nosplit.so; do
    echo "## $w  lib=${lib##*/}"
    PDHG_HIP_LIB=$lib timeout 600 python tools/solve_demo.py $w --verbosity 0 --breakdown 2>&1 | tail -8 | head -5
  done
done
