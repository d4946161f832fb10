#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp in pipeline', d['roofline']['avg_launch_us'], 'alone', d['roofline']['alone']['avg_launch_us'], 'wait', s['host_track_wait_us_per_frame'])"; }
for rep in 1 2 3; do for wl in 4k 1080p; do for lib in video-annotator_amd/lib/libvstab.so tools/dev/libvstab_w4.so; do
  v=$(timeout -k 10 200 python tools/ab_bench.py $lib --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "$wl $(basename $lib) rep$rep: $v"
done; done; done | tee gpurun_out/r04_warp_register_cap_ab.txt
