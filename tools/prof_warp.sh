#!/bin/bash
# usage (GPU box): tools/prof_warp.sh <tag>   -- kernel trace + PMC passes of the fused warp kernel alone (tools/quick_warp_time.py;
# QMODE / QW / QH / QFMT select the instantiation); writes gpurun_out/prof_<tag>/summary.txt.  Counters in passes of their own.
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-warp}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/quick_warp_time.py > $OUT/trace.log 2>&1
i=0
while read -r counters; do
  i=$((i+1))
  rocprofv3 --pmc $counters --output-format csv -d $OUT/pmc$i -- python3 $R/tools/quick_warp_time.py > $OUT/pmc$i.log 2>&1 || echo "pass $i failed: $counters"
done <<'LIST'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_DATA_FIFO_FULL
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_IFETCH_LEVEL SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_THREAD_CYCLES_VALU
FETCH_SIZE
WRITE_SIZE GRBM_GUI_ACTIVE
LIST
python3 $R/tools/summarize_pmc.py $OUT ${KERN:-k_warp_fused}
grep us/frame $OUT/trace.log
