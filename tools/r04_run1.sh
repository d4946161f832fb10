#!/bin/bash
# round 4, first GPU call: the whole -m gpu suite, the tracker timeline with and without the warp beside it, a default bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputest1.log 2>&1; rc=$?
tail -5 gpurun_out/r04_gputest1.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/lk_timeline.py > gpurun_out/r04_lk_timeline_pipeline.txt 2>&1 || { tail -5 gpurun_out/r04_lk_timeline_pipeline.txt; exit 1; }
VSTAB_DEV_SKIP_WARP=1 timeout -k 10 200 python tools/lk_timeline.py > gpurun_out/r04_lk_timeline_no_warp.txt 2>&1 || { tail -5 gpurun_out/r04_lk_timeline_no_warp.txt; exit 1; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench1.json 2> gpurun_out/r04_bench1.err || { tail -5 gpurun_out/r04_bench1.err; exit 1; }
cat gpurun_out/r04_bench1.json | cut -c1-600
