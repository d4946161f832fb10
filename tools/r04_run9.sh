#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -k "unplanned" > gpurun_out/r04_gputest9.log 2>&1; rc=$?
tail -30 gpurun_out/r04_gputest9.log
exit $rc
