#!/bin/bash
export PDHG_DEV=1   # development variables on (csrc/common.hpp: dev_env)
# SQ / TA / TCP / TCC counter passes for the products of ONE matrix shape (default: banded 10M +-50000, where the stream
# kernel's gathers all hit L2): separate `rocprofv3 --kernel-trace --pmc` runs (no other trace domain) of
# tools/shape_table.py --only "<shape>" --no-vendor.  Run on the GPU box:   tools/pmc_stream.sh "<shape title substring>" <out label> [kernel substring]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
SHAPE="${1:-banded 10M +-50000}"
LABEL="${2:-banded50k}"
NEEDLE="${3:-spmv_stream_kernel}"
O=$R/gpurun_out/pmc_stream/$LABEL
rm -rf $O; mkdir -p $O
export SHAPE_CACHE_DIR=/tmp/shape_cache
B="python $R/tools/shape_table.py --only \"$SHAPE\" --no-vendor"
eval $B > $O/plain.txt 2>&1          # fills the cache; the un-profiled timing for reference
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
         "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
         "TA_BUSY_sum TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" \
         "TCC_EA0_RDREQ_sum TCC_BUSY_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum" \
         "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD" ; do
  i=$((i+1))
  eval timeout 400 rocprofv3 --kernel-trace --pmc $C -d $O/p$i -- $B > $O/p$i.log 2>&1 || echo "pass $i failed"
done
eval timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -- $B > $O/kt.log 2>&1 || echo "kernel trace failed"
cd $R; python tools/pmc_summary.py gpurun_out/pmc_stream/$LABEL "$NEEDLE" > gpurun_out/pmc_stream/$LABEL/summary.json
find $O -name "*.db" -size +20M -delete     # the merge back is capped
find $O -name "*kernel_stats.csv" | head -1 | xargs -r head -12
tail -3 $O/plain.txt
