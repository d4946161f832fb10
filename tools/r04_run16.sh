#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
QMODE=5 timeout -k 10 200 python tools/wg_timeline.py > gpurun_out/r04_wg_timeline_persist.txt 2>&1 || { tail -20 gpurun_out/r04_wg_timeline_persist.txt; exit 1; }
VSTAB_WARP_PERSIST=0 QMODE=5 timeout -k 10 200 python tools/wg_timeline.py > gpurun_out/r04_wg_timeline_classic.txt 2>&1 || { tail -20 gpurun_out/r04_wg_timeline_classic.txt; exit 1; }
head -30 gpurun_out/r04_wg_timeline_persist.txt; echo ======; head -8 gpurun_out/r04_wg_timeline_classic.txt
