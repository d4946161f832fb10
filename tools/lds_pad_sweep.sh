#!/bin/bash
# usage (GPU box): tools/lds_pad_sweep.sh -- development build: dwords of padding on the LDS row pitch of the fused warp's staged box;
# kernel time alone (4K, exact map) and the LDS counters (own PMC pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pad in 0 4 8 12 16; do
  export VSTAB_LDS_PAD=$pad QDEV=1
  t=$(python3 $R/tools/quick_warp_time.py 2>&1 | grep warp | sed 's/.*: \([0-9.]* us\).*/\1/')
  O=$R/gpurun_out/prof_pad$pad; rm -rf $O; mkdir -p $O
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $O/pmc1 -- python3 $R/tools/quick_warp_time.py > $O/log 2>&1
  python3 $R/tools/summarize_pmc.py $O k_warp_fused 2>&1 | grep -E "LDS" | sed "s/mean_per_dispatch=//; s/  n=216//" | tr "\n" " "
  echo " <- pad $pad dwords: $t"
  rm -rf $O
done
