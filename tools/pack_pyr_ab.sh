#!/bin/bash
# copy of the frame and first pyramid level in one launch (k_pack_pyr, default) against two launches (VSTAB_PACK_PYR=0): parity tests (the tests' sources are copied), then --ingest copy rates
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_pipeline_gpu.py tests/test_refcl_gpu.py tests/test_lens_gpu.py -m gpu -x -q > gpurun_out/r04_t40.log 2>&1 || { tail -30 gpurun_out/r04_t40.log; exit 1; }
tail -2 gpurun_out/r04_t40.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'track wait', s['host_track_wait_us_per_frame'])"; }
for wl in 4k 1080p; do for rep in 1 2 3; do for fuse in 1 0; do
  v=$(VSTAB_PACK_PYR=$fuse timeout -k 10 300 python bench.py --workload $wl --ingest copy --steps 40 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "$wl --ingest copy, copy + level 1 in one launch=$fuse rep$rep: $v"
done; done; done | tee gpurun_out/r04_pack_pyr_ab.txt
