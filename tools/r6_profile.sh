#!/bin/bash
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
# Round-6 measurement set (run on the GPU box): the driver's command checked the way the driver reads it, rocprofv3 kernel
# stats of the same commands, PMC traffic, the N = 8 command over the test transport.  Outputs under gpurun_out/r6prof/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6prof
mkdir -p $O
cd $R
# 1. the default command (what the driver runs): compact line + bench_details.json
bash tools/r6_bench_check.sh --steps 20 --warmup 5 | tail -3
cp gpurun_out/r6/bench_stdout.txt $O/r06_bench_default_line.json
cp gpurun_out/r6/bench_details.json $O/r06_bench_default.json
cd /tmp && export TMPDIR=/tmp
# 2. the same command under rocprofv3 (full record as the line: tools/rocprof_summary.py reads its `kernels`)
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_default -- python $R/bench.py --no-cpu-baseline --no-self-profile --no-vendor --full-line --no-details > $O/bench_default_under_rocprof.json 2> $O/prof_default.log
python $R/tools/rocprof_summary.py $O/prof_default $O/bench_default_under_rocprof.json > $O/r06_rocprof_summary.json
for WL in pagerank l1svm; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/profplain_$WL -- python $R/bench.py --workload $WL --plain-launches --steps 500 --warmup 50 --no-cpu-baseline --no-other-configs --no-self-profile --no-vendor --full-line --no-details > $O/bench_${WL}_plain_under_rocprof.json 2> $O/profplain_$WL.log
  python $R/tools/rocprof_summary.py $O/profplain_$WL $O/bench_${WL}_plain_under_rocprof.json > $O/r06_${WL}_rocprof_summary.json
done
# 3. PMC traffic
cd $R
bash tools/pmc_traffic.sh "round 6" random pagerank l1svm > $O/pmc.log 2>&1
cp gpurun_out/pmc_traffic/pmc_traffic.json $O/ 2>/dev/null
rm -rf $O/prof_default $O/profplain_pagerank $O/profplain_l1svm gpurun_out/pmc_traffic/*/p1 gpurun_out/pmc_traffic/*/p2
# 4. the driver's N = 8 command over the test-only stand-in transport (one GPU), exchange overlapped: a functional run
export HSA_ENABLE_IPC_MODE_LEGACY=0
PDHG_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29811 bench.py --gpus 8 --steps 20 --warmup 5 --dist-overlap > $O/r06_bench_fake_rccl_8ranks_overlap_line.json 2> $O/bench_fake8.err
cp bench_details.json $O/r06_bench_fake_rccl_8ranks_overlap.json 2>/dev/null
tail -1 $O/r06_bench_fake_rccl_8ranks_overlap_line.json | cut -c1-600
tail -3 $O/bench_fake8.err | cut -c1-300
ls -la $O
