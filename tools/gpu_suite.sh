#!/bin/bash
# whole -m gpu suite (one process), then the default bench lines
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputest_s2.log 2>&1; rc=$?
tail -5 gpurun_out/r04_gputest_s2.log
exit $rc
