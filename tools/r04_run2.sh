#!/bin/bash
# round 4: the segment tracker -- the -m gpu suite, then A/B of frames per tracker launch at 4K and 1080p (same box, alternating)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputest2.log 2>&1; rc=$?
tail -5 gpurun_out/r04_gputest2.log
[ $rc -eq 0 ] || exit $rc
out=gpurun_out/r04_segment_ab.txt; : > $out
for rep in 1 2; do
 for wl in 4k 1080p; do
  for seg in 1 2 4 8; do
    v=$(VSTAB_LK_SEGMENT=$seg VSTAB_DEBUG_SPEC=${DBG:-} timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass 2>gpurun_out/r04_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['parity_check'], d['roofline']['avg_launch_us'], d['stages_timed_region']['host_track_wait_us_per_frame'])") || { tail -5 gpurun_out/r04_ab.err; exit 1; }
    echo "$wl segment<=$seg rep$rep: $v" | tee -a $out
  done
 done
done
