#!/bin/bash
# tracker change: parity tests of the tracker and the pipeline, then the timeline and bench lines
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_track_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q > gpurun_out/r04_t27.log 2>&1 || { tail -30 gpurun_out/r04_t27.log; exit 1; }
tail -3 gpurun_out/r04_t27.log
STEPS=20 bash tools/r04_tl.sh
