#!/bin/bash
# segment-target / read-ahead sweep with the tracker that fetches ahead inside a segment
set -o pipefail
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'wait', s['host_track_wait_us_per_frame'])"; }
for wl in 1080p 4k; do for rep in 1 2; do for cfg in "4 8" "6 8" "6 12" "8 12" "8 16"; do
  set -- $cfg
  v=$(VSTAB_LK_SEG_TARGET=$1 VSTAB_PREFETCH=$2 timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "$wl seg_target=$1 prefetch=$2 rep$rep: $v"
done; done; done | tee gpurun_out/r04_seg_sweep2.txt
