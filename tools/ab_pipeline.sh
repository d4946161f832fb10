#!/bin/bash
# usage (GPU box): [ROUNDS=3] tools/ab_pipeline.sh lib.so...  -- same-box A/B of library builds on the 4K pipeline, rounds interleaved
for round in $(seq 1 ${ROUNDS:-3}); do
  for lib in "$@"; do
    timeout -k 10 300 python tools/ab_bench.py $lib --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass $BENCH_ARGS 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stages']; print('round $round $(basename $lib):', d['value'], 'fps  warp in pipeline', d['roofline']['avg_launch_us'], 'alone', d['roofline']['alone']['avg_launch_us'], ' stage table: lk', s['gpu_lk_us_per_frame'], 'pyr', s['gpu_pyramid_us_per_frame'], 'wait', s['host_track_wait_us_per_frame'], d['parity_check'])"
  done
done
