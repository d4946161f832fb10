#!/bin/bash
# usage (GPU box): tools/prof_ablate.sh  -- VALU instruction count / active cycles / kernel time of the dev build under each ablation
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export QDEV=1
for a in 0 2 4 8 14; do
  OUT=$R/gpurun_out/prof_abl$a; mkdir -p $OUT
  VSTAB_ABLATE=$a rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/quick_warp_time.py > $OUT/trace.log 2>&1
  VSTAB_ABLATE=$a rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -- python3 $R/tools/quick_warp_time.py > $OUT/pmc1.log 2>&1
  echo "== ablate $a"; python3 $R/tools/summarize_pmc.py $OUT k_warp_fused
done
