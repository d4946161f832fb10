#!/bin/bash
# the host's wait for corners at key frames against the read-ahead depth (VSTAB_PREFETCH), 4K and config 5
set -o pipefail
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'corners wait', s['host_corners_us_per_frame'], 'track wait', s['host_track_wait_us_per_frame'])"; }
for wl in 4k 4k-p010; do for rep in 1 2 3; do for pf in 8 12 16; do
  v=$(VSTAB_PREFETCH=$pf timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "$wl prefetch=$pf rep$rep: $v"
done; done; done | tee gpurun_out/r04_keyframe_wait.txt
