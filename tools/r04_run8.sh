#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'wait', s['host_track_wait_us_per_frame'])"; }
for rep in 1 2; do for wl in 1080p 4k; do for cfg in "8 4" "12 4" "12 6" "16 8" "16 4"; do
  set -- $cfg
  v=$(VSTAB_PREFETCH=$1 VSTAB_LK_SEG_TARGET=$2 timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "$wl prefetch=$1 seg_target=$2 rep$rep: $v"
done; done; done | tee gpurun_out/r04_prefetch_sweep.txt
