#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_track_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q -k "not four_ranks and not rccl and not launches_its_own" > gpurun_out/r04_gputest7.log 2>&1; rc=$?
tail -4 gpurun_out/r04_gputest7.log
[ $rc -eq 0 ] || exit $rc
bash tools/r04_tl.sh
