#!/bin/bash
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
# ThreadSanitizer run of the shard pool (one issuing host thread per shard) on the GPU box: 4 shards on one device,
# peer-kernel exchange, 80 adaptive steps, getters.  The HIP runtime and the interpreter are not instrumented, so reports
# whose racing accesses lie inside libamdhip64 / libhsa-runtime are noise of the method; tools/tsan_classify.py counts the
# reports with a racing access inside the library itself (the figure that matters).
cd "$GRAFT_REPO_ROOT"
LIB=firstorderlp.jl_amd/csrc/libpdhg_hip_tsan.so
[ -f $LIB ] || python -c "import folp_loader; p=folp_loader.load(); p._lib.build_sanitized('tsan', verbose=True)"
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.tsan-x86_64.so)
OUT=gpurun_out/r4_tsan_shards.txt
rm -f gpurun_out/tsan_report.*
echo "# $LIB under $RT, tools/tsan_drive.py 4" > $OUT
LD_PRELOAD=$RT TSAN_OPTIONS="report_signal_unsafe=0 history_size=4 log_path=gpurun_out/tsan_report exitcode=0" PDHG_HIP_LIB=$LIB \
  timeout 600 python tools/tsan_drive.py 4 2>&1 | grep -v "^$" | tail -5 >> $OUT
python tools/tsan_classify.py gpurun_out/tsan_report.* >> $OUT 2>&1
cat $OUT
