#!/bin/bash
# round 4: estimate pipelined one frame deep -- suite, then frames/s at 4K and 1080p with the frame loop in Python and in C
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputest4.log 2>&1; rc=$?
tail -5 gpurun_out/r04_gputest4.log
[ $rc -eq 0 ] || exit $rc
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'wait', s['host_track_wait_us_per_frame'], 'est', s['host_estimate_us_per_frame'])"; }
out=gpurun_out/r04_estimate_pipelined.txt; : > $out
for rep in 1 2 3; do
 for wl in 4k 1080p; do
  for pull in single batch; do
    v=$(timeout -k 10 200 python bench.py --workload $wl --pull $pull --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
    echo "$wl pull=$pull rep$rep: $v" | tee -a $out
  done
 done
done
