#!/bin/bash
# warp instruction trims: parity (warp / refcl / p010 / lens tests), then the 4K bench lines (pipeline, warp alone, config 5)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_warp_gpu.py tests/test_refcl_gpu.py tests/test_p010_gpu.py tests/test_lens_gpu.py -m gpu -x -q > gpurun_out/r04_t15.log 2>&1 || { tail -30 gpurun_out/r04_t15.log; exit 1; }
tail -2 gpurun_out/r04_t15.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['parity_check'], 'warp', r['avg_launch_us'], 'alone', (r.get('alone') or {}).get('avg_launch_us'))"; }
for rep in 1 2 3; do
  v=$(timeout -k 10 200 python bench.py --workload 4k --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "4k rep$rep: $v"
done
v=$(timeout -k 10 200 python bench.py --workload 4k-p010 --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
echo "config5: $v"
v=$(timeout -k 10 200 python bench.py --workload 1080p --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
echo "1080p: $v"
