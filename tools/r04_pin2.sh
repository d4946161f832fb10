#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'wait', s['host_track_wait_us_per_frame'])"; }
for rep in 1 2 3; do for set in 64-127 0-63 0-7 0-7,128-135 8-15 32-39 96-103; do
  v=$(VSTAB_BENCH_PIN=0 timeout -k 10 200 taskset -c $set python bench.py --workload 1080p --steps 60 --warmup 5 --no-cpu-baseline --skip-copy-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "1080p cpus=$set rep$rep: $v"
done; done | tee gpurun_out/r04_pin2.txt
