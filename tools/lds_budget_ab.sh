#!/bin/bash
# LDS budget of a warp tile in the running 4K pipeline (development build, VSTAB_LDS_KB): does a tracker workgroup (36 KB) beside FOUR tiles pay now that the warp is the limiter?
set -o pipefail
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'alone', d['roofline']['alone']['avg_launch_us'], 'wait', s['host_track_wait_us_per_frame'])"; }
for rep in 1 2 3; do for kb in 40 36 31 30 28 26; do
  v=$(VSTAB_LDS_KB=$kb timeout -k 10 200 python tools/ab_bench.py tools/dev/libvstab_dev.so --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "4k warp LDS budget $kb KB rep$rep: $v"
done; done | tee gpurun_out/r04_lds_budget_ab2.txt
