#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'corners wait', s['host_corners_us_per_frame'], 'track wait', s['host_track_wait_us_per_frame'])"; }
for rep in 1 2 3; do for cfg in "" "--ingest copy"; do
  v=$(timeout -k 10 300 python bench.py --workload 4k-p010 $cfg --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "4k-p010 [$cfg] rep$rep: $v"
done; done | tee gpurun_out/r04_cfg5.txt
timeout -k 10 600 python -m pytest tests/test_p010_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q -k "10bit_pixels or p010_input or hold" 2>&1 | tail -2
