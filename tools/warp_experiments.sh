#!/bin/bash
# usage (GPU box): tools/warp_experiments.sh  -- timing of the fused warp kernel under the development switches
export QDEV=1
echo "== baseline (product build)"; QDEV= python tools/quick_warp_time.py 2>&1 | grep warp
echo "== mode 5 (product build)"; QDEV= QMODE=5 python tools/quick_warp_time.py 2>&1 | grep warp
for kb in 40 32 26; do for t in 0.5 1; do echo "== lds $kb KB tail rounds $t"; VSTAB_LDS_KB=$kb VSTAB_TAIL_ROUNDS=$t python tools/quick_warp_time.py 2>&1 | grep warp; done; done
echo "== 1080p (product)"; QDEV= QW=1920 QH=1080 python tools/quick_warp_time.py 2>&1 | grep warp
for kb in 24 20 16; do echo "== 1080p rows 4 lds $kb"; QW=1920 QH=1080 VSTAB_LDS_KB=$kb python tools/quick_warp_time.py 2>&1 | grep warp; done
for kb in 40 32 26; do echo "== 1080p rows 8 lds $kb"; QW=1920 QH=1080 VSTAB_ROWS=8 VSTAB_LDS_KB=$kb python tools/quick_warp_time.py 2>&1 | grep warp; done
