#!/bin/bash
# usage (GPU box): tools/ab_variants.sh lib.so... -- same-box A/B of library builds: the fused warp alone (4K exact map, 1080p) and the
# 4K pipeline, two rounds interleaved so that drift of the box shows up as a difference between the rounds
for round in 1 2; do
  for lib in "$@"; do
    a=$(QDEV=$lib python tools/quick_warp_time.py 2>&1 | grep warp | sed 's/.*: \([0-9.]* us\).*/\1/')
    b=$(QDEV=$lib QW=1920 QH=1080 python tools/quick_warp_time.py 2>&1 | grep warp | sed 's/.*: \([0-9.]* us\).*/\1/')
    c=$(timeout -k 10 300 python tools/ab_bench.py $lib --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'fps, warp in pipeline', d['roofline']['avg_launch_us'], 'alone', d['roofline']['alone']['avg_launch_us'], d['parity_check'])")
    echo "round $round $(basename $lib): 4K alone $a  1080p alone $b  pipeline $c"
  done
done
