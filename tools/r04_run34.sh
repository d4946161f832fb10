#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'wait', s['host_track_wait_us_per_frame'])"; }
for wl in 1080p 4k; do for rep in 1 2 3; do for cfg in single batch; do
  v=$(timeout -k 10 200 python bench.py --workload $wl --pull $cfg --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "$wl pull=$cfg rep$rep: $v"
done; done; done | tee gpurun_out/r04_pull_batch_ab.txt
