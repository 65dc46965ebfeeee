mkdir -p gpurun_out/r3solve
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
O=gpurun_out/r3solve/r03_solve_demo.txt
: > $O
for args in "--workload random --n 10000000 --iteration_limit 60000" "--workload random --n 1000000" "--workload pagerank --n 1000000" "--workload l1svm --iteration_limit 40000"; do
  timeout 900 python tools/solve_demo.py $args --verbosity 0 2>/dev/null | tail -2 >> $O
done
for ord in strict; do
  echo "# PDHG_ROW_ORDER=$ord" >> $O
  for args in "--workload pagerank --n 1000000" "--workload l1svm --iteration_limit 40000"; do
    PDHG_ROW_ORDER=$ord timeout 900 python tools/solve_demo.py $args --verbosity 0 2>/dev/null | tail -2 >> $O
  done
done
cat $O
