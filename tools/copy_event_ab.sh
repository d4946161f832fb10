#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_p010_gpu.py -m gpu -x -q > gpurun_out/r04_t39.log 2>&1 || { tail -30 gpurun_out/r04_t39.log; exit 1; }
tail -2 gpurun_out/r04_t39.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'wait', s['host_track_wait_us_per_frame'])"; }
for rep in 1 2 3; do
  v=$(timeout -k 10 200 python bench.py --workload 4k-p010 --ingest copy --steps 40 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "4k-p010 --ingest copy rep$rep: $v"
done | tee gpurun_out/r04_copy_event_p010.txt
