#!/bin/bash
# run-to-run spread of the 4K line against the read-ahead depth and the detection's stream (how often does a key-frame corner wait show?)
set -o pipefail
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'corners wait', s['host_corners_us_per_frame'])"; }
for rep in 1 2 3 4 5 6; do for cfg in "8 1" "12 1" "8 0" "12 0"; do
  set -- $cfg
  v=$(VSTAB_PREFETCH=$1 VSTAB_DETECT_STREAM=$2 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "prefetch=$1 detection_stream=$2 rep$rep: $v"
done; done | tee gpurun_out/r04_robustness_runs.txt
