#!/bin/bash
# streams in the process against the runtime's four hardware queues: the caller on the default stream or on a stream of its own (the default stream then stays: a fifth),
# the speculative detection on a stream of its own or on the read-ahead stream (VSTAB_DETECT_STREAM=0)
set -o pipefail
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'corners wait', s['host_corners_us_per_frame'], 'track wait', s['host_track_wait_us_per_frame'])"; }
for wl in 4k 1080p; do for rep in 1 2; do for own in 0 1; do for det in 1 0; do
  if [ $own = 1 ]; then export VSTAB_BENCH_OWN_STREAM=1; else unset VSTAB_BENCH_OWN_STREAM; fi
  v=$(VSTAB_DETECT_STREAM=$det timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "$wl caller_own_stream=$own detection_stream=$det rep$rep: $v"
done; done; done; done | tee gpurun_out/r04_stream_count_ab.txt
