#!/bin/bash
# kernel trace of the 4K / 1080p bench command -> tools/warp_overlap_regression.py
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wl in 4k 1080p; do
  OUT=$R/gpurun_out/prof_ovl_$wl
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload $wl --steps 40 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
  echo "== $wl: $(tail -1 $OUT/trace.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'frames/s under the tracer')")"
  python3 $R/tools/warp_overlap_regression.py $OUT
  rm -rf $OUT/trace
done 2>&1 | tee $R/gpurun_out/r04_warp_overlap_regression.txt
