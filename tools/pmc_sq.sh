set -x
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_sq
mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1
B="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --profile-steps 0"
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
         "TA_BUSY_sum TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_BUSY_sum TCC_EA0_RDREQ_DRAM_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/p$i -- $B > $O/p$i.log 2>&1 || echo "pass $i failed"
done
cd $R; python tools/pmc_summary.py gpurun_out/pmc_sq spmv_tiled > gpurun_out/pmc_sq/summary.json; tail -5 gpurun_out/pmc_sq/p1.log
