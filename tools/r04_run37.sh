#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_track_gpu.py -m gpu -x -q -k "pyr_down" > gpurun_out/r04_t36.log 2>&1 || { tail -30 gpurun_out/r04_t36.log; exit 1; }
tail -2 gpurun_out/r04_t36.log
python tools/time_pyr.py 2>&1 | tail -2
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'wait', s['host_track_wait_us_per_frame'])"; }
for wl in 1080p 4k; do for rep in 1 2 3; do
  v=$(timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "$wl rep$rep: $v"
done; done | tee gpurun_out/r04_bench_latest.txt
