bash tools/r6_profile.sh 2>&1 | tail -40
export PDHG_DEV=1
PDHG_SJ_WIDE=0 bash tools/pmc_stream.sh "banded 10M +-50000" sj_narrow spmv_sj_kernel 2>&1 | tail -4
PDHG_SJ_WIDE=1 bash tools/pmc_stream.sh "banded 10M +-50000" sj_wide spmv_sj_kernel 2>&1 | tail -4
mkdir -p gpurun_out/r6prof; cp gpurun_out/pmc_stream/sj_narrow/summary.json gpurun_out/r6prof/r06_sj_kernel_pmc_narrow.json; cp gpurun_out/pmc_stream/sj_wide/summary.json gpurun_out/r6prof/r06_sj_kernel_pmc_wide.json
rm -rf gpurun_out/pmc_stream/*/p* gpurun_out/pmc_stream/*/kt
