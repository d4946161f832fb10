#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'corners wait', s['host_corners_us_per_frame'], 'track wait', s['host_track_wait_us_per_frame'])"; }
for rep in 1 2 3 4 5 6 7 8; do for cfg in "" "--force-dist --dist-backend nccl" "--workload 1080p"; do
  v=$(timeout -k 10 300 python bench.py --gpus 1 $cfg --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "[$cfg] rep$rep: $v"
done; done | tee gpurun_out/r04_robustness_runs2.txt
