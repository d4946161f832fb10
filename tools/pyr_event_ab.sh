#!/bin/bash
# ingest event bound to the pyramid's last kernel (default) against a hipEventRecord behind it (VSTAB_PYR_EVENT_RECORD=1)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q > gpurun_out/r04_t33.log 2>&1 || { tail -30 gpurun_out/r04_t33.log; exit 1; }
tail -2 gpurun_out/r04_t33.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'wait', s['host_track_wait_us_per_frame'])"; }
for wl in 1080p 4k; do for rep in 1 2 3; do for cfg in none VSTAB_PYR_EVENT_RECORD; do
  v=$(env $cfg=1 timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "$wl $cfg rep$rep: $v"
done; done; done | tee gpurun_out/r04_pyr_event_ab.txt
