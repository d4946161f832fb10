#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'wait', s['host_track_wait_us_per_frame'], d['rank_cpus'])"; }
nproc; lscpu | grep -i "numa\|model name\|socket" | head -12
for rep in 1 2 3 4; do for pin in 1 0; do
  v=$(VSTAB_BENCH_PIN=$pin timeout -k 10 200 python bench.py --workload 1080p --steps 60 --warmup 5 --no-cpu-baseline --skip-copy-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "1080p pin=$pin rep$rep: $v"
done; done | tee gpurun_out/r04_pin.txt
