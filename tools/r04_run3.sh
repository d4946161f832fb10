#!/bin/bash
# round 4: fused pyramid levels (A/B against one launch per level), pipeline statistics, warp LDS budget beside the segment tracker
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputest3.log 2>&1; rc=$?
tail -5 gpurun_out/r04_gputest3.log
[ $rc -eq 0 ] || exit $rc
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'wait', d['stages_timed_region']['host_track_wait_us_per_frame'], 'pyr', s['gpu_pyramid_us_per_frame'], 'lk', s['gpu_lk_us_per_frame'])"; }
out=gpurun_out/r04_pyramid_ab.txt; : > $out
for rep in 1 2; do
 for wl in 4k 1080p; do
  for single in "" 1; do
    v=$(VSTAB_PYR_SINGLE=$single timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
    echo "$wl one-launch-per-level=${single:-0} rep$rep: $v" | tee -a $out
  done
 done
done
VSTAB_DEBUG_SPEC=1 VSTAB_LK_CLOCK=1 VSTAB_HOST_TIMING=1 timeout -k 10 200 python bench.py --workload 4k --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass > gpurun_out/r04_stats_4k.json 2> gpurun_out/r04_stats_4k.err || exit 1
grep -v "async selection\|spec launch\|key frame at" gpurun_out/r04_stats_4k.err | tail -25
VSTAB_DEBUG_SPEC=1 VSTAB_LK_CLOCK=1 VSTAB_HOST_TIMING=1 timeout -k 10 200 python bench.py --workload 1080p --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass > gpurun_out/r04_stats_1080p.json 2> gpurun_out/r04_stats_1080p.err || exit 1
grep -v "async selection\|spec launch\|key frame at" gpurun_out/r04_stats_1080p.err | tail -25
out=gpurun_out/r04_warp_lds_ab.txt; : > $out
for rep in 1 2; do
  for kb in 40 36 32 30 26; do
    v=$(VSTAB_LDS_KB=$kb timeout -k 10 200 python tools/ab_bench.py tools/dev/libvstab_dev.so --workload 4k --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
    echo "4k warp LDS budget ${kb} KB rep$rep: $v" | tee -a $out
  done
done
