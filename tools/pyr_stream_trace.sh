#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wl in 1080p 4k; do
  OUT=$R/gpurun_out/prof_pyrgap_$wl
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
  echo "$wl: $(python3 $R/tools/pyr_stream_gaps.py $OUT)"
  python3 $R/tools/analyze_trace.py $OUT 2>/dev/null | head -12
  rm -rf $OUT/trace
done 2>&1 | tee $R/gpurun_out/r04_pyr_stream_gaps.txt
