#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], 'warp', d['roofline']['avg_launch_us'], 'wait', s['host_track_wait_us_per_frame'], 'corners', s['host_corners_us_per_frame'])"; }
for rep in 1 2; do for st in 0 50 100 200 400; do
  x=$(VSTAB_BENCH_STALL_US=$st timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "stall ${st} us every 20th frame, rep$rep: $x"
done; done
