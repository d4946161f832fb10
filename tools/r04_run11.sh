#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_p010_gpu.py tests/test_refcl_gpu.py -m gpu -x -q > gpurun_out/r04_gputest11.log 2>&1; rc=$?
tail -12 gpurun_out/r04_gputest11.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/quick_p010_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_p010_times.txt
