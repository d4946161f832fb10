#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_track_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q > gpurun_out/r04_t35.log 2>&1 || { tail -30 gpurun_out/r04_t35.log; exit 1; }
tail -2 gpurun_out/r04_t35.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'wait', s['host_track_wait_us_per_frame'])"; }
for wl in 1080p 4k; do for rep in 1 2 3; do
  v=$(timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "$wl rep$rep: $v"
done; done | tee gpurun_out/r04_bench_latest.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wl in 1080p 4k; do
  OUT=$R/gpurun_out/prof_pyrgap_$wl
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
  echo "$wl: $(python3 $R/tools/pyr_stream_gaps.py $OUT)"
  python3 $R/tools/analyze_trace.py $OUT 2>/dev/null | head -9
  rm -rf $OUT/trace
done 2>&1 | tee $R/gpurun_out/r04_pyr_stream_gaps2.txt
