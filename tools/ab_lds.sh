#!/bin/bash
# usage (GPU box): tools/ab_lds.sh KB...   -- pipeline bench with the development build and the warp kernel's LDS budget set to each value
for kb in "$@"; do
  VSTAB_LDS_KB=$kb VSTAB_LIB_PATH=video-annotator_amd/lib/libvstab_dev.so timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lds $kb KB:', d['value'], 'fps; warp in pipeline', d['roofline']['avg_launch_us'], 'us, alone', d['roofline']['alone']['avg_launch_us'], 'us; lk', d['stages']['gpu_lk_us_per_frame'], d['parity_check'])"
done
