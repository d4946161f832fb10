#!/bin/bash
set -o pipefail
for rep in 1 2; do for m in 5 0; do for f in 0 1; do
echo "== rs mode $m fmt $f old (72 registers)"; QRS=1 QDEV=tools/dev/libvstab_old.so QMODE=$m QFMT=$f timeout -k 10 120 python tools/quick_warp_time.py 2>&1 | grep warp
echo "== rs mode $m fmt $f new (80 registers)"; QRS=1 QMODE=$m QFMT=$f timeout -k 10 120 python tools/quick_warp_time.py 2>&1 | grep warp
done; done; done
timeout -k 10 600 python -m pytest tests/test_warp_gpu.py tests/test_refcl_gpu.py tests/test_lens_gpu.py -m gpu -x -q 2>&1 | tail -2
