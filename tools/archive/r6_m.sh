export PDHG_DEV=1
for rep in 1 2 3; do for env in "PDHG_DEV=1" "PDHG_STREAM_PIPE=0"; do
  echo "## $env"; env $env timeout 600 python tools/solve_demo.py --workload l1svm --iteration_limit 40000 --verbosity 0 --breakdown 2>&1 | grep -E "run_restart|update_obj|iteration_stats|take_steps"
done; done
