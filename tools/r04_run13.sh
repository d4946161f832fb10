#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'alone', d['roofline']['alone']['avg_launch_us'])"; }
for rep in 1 2; do for cfg in "" "VSTAB_TAIL_ROUNDS=0" "VSTAB_TAIL_ROUNDS=0.25" "VSTAB_TAIL_ROUNDS=1" "VSTAB_ROWS=4 VSTAB_LDS_KB=20" "VSTAB_ROWS=4 VSTAB_LDS_KB=24"; do
  v=$(env $cfg timeout -k 10 200 python tools/ab_bench.py tools/dev/libvstab_dev.so --workload 4k --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "4k dev [$cfg] rep$rep: $v"
done; done | tee gpurun_out/r04_warp_tile_schedule_in_pipeline.txt
