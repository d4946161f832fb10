#!/bin/bash
set -o pipefail
for rep in 1 2 3; do
echo "== map group 2, tap group 4 (product)"; QMODE=5 timeout -k 10 120 python tools/quick_warp_time.py 2>&1 | grep warp
for v in 28 44 48; do echo "== map group ${v:0:1}, tap group ${v:1:1}"; QDEV=tools/dev/libvstab_v$v.so QMODE=5 timeout -k 10 120 python tools/quick_warp_time.py 2>&1 | grep warp; done
done
