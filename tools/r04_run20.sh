#!/bin/bash
# A = 72-register budget for the 64x32-tile kernels (ten scratch instructions in rare paths), B = 80 registers (none)
set -o pipefail
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['parity_check'], 'warp', r['avg_launch_us'], 'alone', (r.get('alone') or {}).get('avg_launch_us'))"; }
for rep in 1 2 3; do for v in A B; do
  cp tools/dev/libvstab_$v.so video-annotator_amd/lib/libvstab.so
  x=$(timeout -k 10 200 python bench.py --workload 4k --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "4k $v rep$rep: $x"
done; done
for v in A B; do
  cp tools/dev/libvstab_$v.so video-annotator_amd/lib/libvstab.so
  x=$(timeout -k 10 200 python bench.py --workload 4k --out-format nv12 --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "4k nv12 $v: $x"
  x=$(timeout -k 10 200 python bench.py --workload 4k --map-precision ieee --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "4k ieee $v: $x"
done
