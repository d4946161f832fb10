#!/bin/bash
# usage (GPU box): tools/lds_sweep.sh -- the 4K pipeline over the development build with the fused warp's LDS budget
# swept (VSTAB_LDS_KB): does a smaller tile budget let a tracker workgroup (21 KB) sit beside FOUR warp workgroups?
for kb in ${KBS:-40 36 34 32 28}; do
  VSTAB_LDS_KB=$kb timeout -k 10 300 python tools/ab_bench.py tools/dev/libvstab_dev.so --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stages']; print('lds_kb=$kb:', d['value'], 'fps  warp in pipeline', d['roofline']['avg_launch_us'], 'alone', d['roofline']['alone']['avg_launch_us'], ' lk', s['gpu_lk_us_per_frame'], 'pyr', s['gpu_pyramid_us_per_frame'], d['parity_check'])"
done
