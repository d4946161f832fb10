#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_track_gpu.py -m gpu -x -q > gpurun_out/r04_gputest12.log 2>&1; rc=$?
tail -4 gpurun_out/r04_gputest12.log
[ $rc -eq 0 ] || exit $rc
bash tools/r04_prof.sh
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'wait', s['host_track_wait_us_per_frame'])"; }
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do v=$(timeout -k 10 200 python bench.py --workload 4k --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass 2>gpurun_out/r04_ab.err | line); echo "4k rep$rep: $v"; done
rm -rf gpurun_out/prof_r04_4k/trace gpurun_out/prof_r04_1080p/trace
