#!/bin/bash
# usage (GPU box): [LKS="1 2"] tools/ab_run.sh lib.so... -- bench.py (4K pipeline) over each build; prints fps and the warp / LK stage times
for lib in "$@"; do
  for st in ${LKS:-2}; do
  VSTAB_LK_STREAMS=$st timeout -k 10 300 python tools/ab_bench.py $lib --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stages']; print('$lib lk_streams=$st:', d['value'], 'fps  warp in pipeline', d['roofline']['avg_launch_us'], 'alone', d['roofline']['alone']['avg_launch_us'], ' stage table: lk', s['gpu_lk_us_per_frame'], 'pyr', s['gpu_pyramid_us_per_frame'], 'wait', s['host_track_wait_us_per_frame'], d['parity_check'])"
  done
done
