#!/bin/bash
# A/B on one box: warp launches with hipExtAnyOrderLaunch (VSTAB_EXP_WARP_ANY_ORDER=1) against stream-ordered ones
set -o pipefail
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], 'warp', d['roofline']['avg_launch_us'], 'alone', d['roofline']['alone']['avg_launch_us'], 'wait', s['host_track_wait_us_per_frame'], d['parity_check'])"; }
for wl in 4k 1080p; do for rep in 1 2 3; do for ao in 0 1; do
  if [ $ao = 1 ]; then export VSTAB_EXP_WARP_ANY_ORDER=1; else unset VSTAB_EXP_WARP_ANY_ORDER; fi
  x=$(timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "$wl any_order=$ao rep$rep: $x"
done; done; done 2>&1 | tee gpurun_out/r04_any_order_ab.txt
