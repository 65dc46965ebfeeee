# round 6: the stream kernels after the waitcnt fixes (unsigned column offsets; one explicit wait before the gathers; the
# long chunk's gathers ahead of its products) against r5's tree, on the latency-bound LPs
export PDHG_DEV=1 SHAPE_CACHE_DIR=/tmp/shapes
mkdir -p gpurun_out/r6 /tmp/shapes
T=gpurun_out/r6/stream_waitcnt.txt
: > $T
runhere() { echo "## $1 | current | env: $2" >> $T; env $2 timeout 900 python tools/shape_table.py --only "$1" $3 2>&1 | grep -v "^#" >> $T; }
runr5() { test -d .r5tree && { echo "## $1 | r5 tree | env: $2" >> $T; (cd .r5tree && env $2 timeout 900 python tools/shape_table.py --only "$1" --no-vendor 2>&1 | grep -v "^#") >> $T; }; }
for shape in "pagerank 1M" "l1svm" "4M-pagerank" "10M-banded +-50000, lognormal rows" "1M-lognormal rows"; do
  runhere "$shape" "" "--no-vendor"
  runr5 "$shape" ""
done
runhere "banded 10M +-50000" "PDHG_SJ=0" "--no-vendor"
runr5 "banded 10M +-50000" "PDHG_SJ=0"
cat $T | cut -c1-250
for wl in l1svm pagerank; do
  echo "## bench --workload $wl (current)"; python bench.py --workload $wl --steps 4000 --warmup 300 --no-cpu-baseline --no-self-profile --no-vendor --no-details 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'])"
  test -d .r5tree && { echo "## bench --workload $wl (r5 tree)"; (cd .r5tree && python bench.py --workload $wl --steps 4000 --warmup 300 --no-cpu-baseline --no-self-profile --no-vendor 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'])"); }
done 2>&1 | tee -a $T
