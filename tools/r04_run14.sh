#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'wait', s['host_track_wait_us_per_frame'], 'corners', s['host_corners_us_per_frame'], 'est', s['host_estimate_us_per_frame'])"; }
for rep in 1 2; do for pf in 8 12 16; do
  v=$(VSTAB_PREFETCH=$pf VSTAB_DEBUG_SPEC=1 timeout -k 10 200 python bench.py --workload 4k --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "4k prefetch=$pf rep$rep: $v | $(grep 'tracker launches' gpurun_out/r04_ab.err | head -1 | sed 's/.*key frames pre-launched/key frames pre-launched/')"
done; done
