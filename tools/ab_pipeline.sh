#!/bin/bash
# usage (GPU box): tools/ab_pipeline.sh lib1.so lib2.so ...   -- pipeline bench (4K, 20 x 64 frames) with each build of the library
for lib in "$@"; do
  VSTAB_LIB_PATH=video-annotator_amd/lib/$lib timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], 'fps; warp in pipeline', d['roofline']['avg_launch_us'], 'us, alone', d['roofline']['alone']['avg_launch_us'], 'us;', d['parity_check'])"
done
