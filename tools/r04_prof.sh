#!/bin/bash
# rocprofv3 kernel trace of the bench command (4K and 1080p), summaries into gpurun_out/
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wl in 4k 1080p; do
  OUT=$R/gpurun_out/prof_r04_$wl
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass > $OUT/bench.json 2> $OUT/trace.log || { tail -5 $OUT/trace.log; exit 1; }
  f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
  cp $f $R/gpurun_out/r04_bench_pipeline_${wl}_kernel_stats.csv
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if 'vstab::' in r['Name']:
        print(f"{r['Name'][:95]:95s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_us={float(r['AverageNs'])/1e3:8.2f} min_us={float(r['MinNs'])/1e3:7.2f} max_us={float(r['MaxNs'])/1e3:8.2f} pct={r['Percentage']}")
PY
  tail -1 $OUT/bench.json | cut -c1-300
done
