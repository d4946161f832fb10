#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -k "every_clip_length" --durations=5 > gpurun_out/r04_t25.log 2>&1; rc=$?
tail -25 gpurun_out/r04_t25.log
exit $rc
