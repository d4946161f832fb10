#!/bin/bash
# usage (here, after gpurun merged gpurun_out/): bash tools/publish_profiles.sh <tag> <round-prefix>
# copies the judged summaries of gpurun_out/prof_<tag>/ (written by tools/prof_bench.sh) into profiles/
set -e
TAG=${1:?tag}
PFX=${2:-r03}
SRC=gpurun_out/prof_$TAG
cp $SRC/summary.txt profiles/${PFX}_bench_pipeline_rocprof_summary.txt
cp $SRC/traffic.json profiles/traffic_4k.json
STATS=$(find $SRC/trace -name "*kernel_stats*.csv" 2>/dev/null | head -1)
(head -1 $STATS; grep vstab:: $STATS) > profiles/${PFX}_bench_pipeline_kernel_stats.csv
python3 - "$PFX" <<'PY'
import csv, json, sys
pfx = sys.argv[1]
rows = list(csv.DictReader(open(f"profiles/{pfx}_bench_pipeline_kernel_stats.csv")))
w = [r for r in rows if "k_warp_fused" in r["Name"]][0]
t = json.load(open("profiles/traffic_4k.json"))
t["rocprof_avg_launch_us"] = round(float(w["AverageNs"]) / 1e3, 2)
json.dump(t, open("profiles/traffic_4k.json", "w"), indent=1)
print("k_warp_fused rocprof avg us:", t["rocprof_avg_launch_us"])
PY
