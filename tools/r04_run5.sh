#!/bin/bash
# A/B of library builds on one box: 1080p and 4K pipeline, rounds interleaved; then the tracker timeline (development build)
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/r04_ab_libs.txt; : > $out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'wait', s['host_track_wait_us_per_frame'])"; }
for rep in 1 2 3; do
 for wl in 1080p 4k; do
  for lib in "$@"; do
    v=$(timeout -k 10 200 python tools/ab_bench.py $lib --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
    echo "$wl $(basename $lib) rep$rep: $v" | tee -a $out
  done
 done
done
timeout -k 10 200 python tools/lk_timeline.py 2>&1 | grep -v "^  launch\|amdgpu.ids" | tee gpurun_out/r04_lk_timeline_segments.txt
