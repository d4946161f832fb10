#!/bin/bash
# tile-after-tile workgroups of the fused warp: parity first, then kernel time alone (4K), with and without, two register caps
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_warp_gpu.py tests/test_refcl_gpu.py tests/test_lens_gpu.py -m gpu -x -q > gpurun_out/r04_t17.log 2>&1 || { tail -30 gpurun_out/r04_t17.log; exit 1; }
tail -2 gpurun_out/r04_t17.log
for rep in 1 2; do
echo "== persist, cap 7 (product)";  QMODE=5 timeout -k 10 120 python tools/quick_warp_time.py 2>&1 | grep warp
echo "== classic, cap 7 (product, VSTAB_WARP_PERSIST=0)";  VSTAB_WARP_PERSIST=0 QMODE=5 timeout -k 10 120 python tools/quick_warp_time.py 2>&1 | grep warp
echo "== persist, cap 6";  QDEV=tools/dev/libvstab_w6.so QMODE=5 timeout -k 10 120 python tools/quick_warp_time.py 2>&1 | grep warp
echo "== classic, cap 6";  VSTAB_WARP_PERSIST=0 QDEV=tools/dev/libvstab_w6.so QMODE=5 timeout -k 10 120 python tools/quick_warp_time.py 2>&1 | grep warp
done
