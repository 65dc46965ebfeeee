#!/bin/bash
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
# Round-5 measurement set (run on the GPU box): bench lines, rocprofv3 kernel stats of the same commands,
# PMC traffic.  Outputs under gpurun_out/r5prof/.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the default command (what the driver runs), plain and under rocprofv3
timeout 900 python $R/bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_default -- python $R/bench.py --no-cpu-baseline --no-self-profile > $O/bench_default_under_rocprof.json 2> $O/prof_default.log
python $R/tools/rocprof_summary.py $O/prof_default $O/bench_default_under_rocprof.json > $O/r05_rocprof_summary.json
# 2. configs[2] / configs[3] as headline workloads, under rocprofv3: (a) the command as the driver would time it (one
#    graph launch / one persistent kernel per trial), (b) the same with --plain-launches: every kernel its own launch in
#    stream order, so that each product's kernels add up (no concurrent graph branches, no fused trial kernel)
for WL in pagerank l1svm; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$WL -- python $R/bench.py --workload $WL --steps 500 --warmup 50 --no-cpu-baseline --no-other-configs --no-self-profile > $O/bench_${WL}_under_rocprof.json 2> $O/prof_$WL.log
  python $R/tools/rocprof_summary.py $O/prof_$WL $O/bench_${WL}_under_rocprof.json > $O/r05_${WL}_onelaunch_rocprof_summary.json
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/profplain_$WL -- python $R/bench.py --workload $WL --plain-launches --steps 500 --warmup 50 --no-cpu-baseline --no-other-configs --no-self-profile > $O/bench_${WL}_plain_under_rocprof.json 2> $O/profplain_$WL.log
  python $R/tools/rocprof_summary.py $O/profplain_$WL $O/bench_${WL}_plain_under_rocprof.json > $O/r05_${WL}_rocprof_summary.json
done
# 3. PMC traffic
cd $R
bash tools/pmc_traffic.sh "round 5" random pagerank l1svm > $O/pmc.log 2>&1
cp gpurun_out/pmc_traffic/pmc_traffic.json $O/ 2>/dev/null
rm -rf $O/prof_default $O/prof_pagerank $O/prof_l1svm $O/profplain_pagerank $O/profplain_l1svm gpurun_out/pmc_traffic/*/p1 gpurun_out/pmc_traffic/*/p2
ls -la $O
python -c "
import json
d=json.load(open('$O/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'])
for o in d.get('other_configs', []): print(o['config']['workload'][:40], o.get('value'), o.get('roofline', {}).get('frac'), o.get('roofline', {}).get('kernel'))
"
