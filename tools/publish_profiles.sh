#!/bin/bash
# usage (here, after gpurun merged gpurun_out/): bash tools/publish_profiles.sh <tag> <round-prefix> [name] [kernel]
# copies the judged summaries of gpurun_out/prof_<tag>/ (written by tools/prof_bench.sh) into profiles/; name = bench_pipeline (the
# driver's BGR line, whose traffic file bench.py quotes as committed_profile) or e.g. bench_pipeline_nv12_planar with kernel k_warp_planar
set -e
TAG=${1:?tag}
PFX=${2:-r05}
NAME=${3:-bench_pipeline}
KERN=${4:-k_warp_fused}
SRC=gpurun_out/prof_$TAG
TRAFFIC=profiles/traffic_4k.json
[ "$NAME" = bench_pipeline ] || TRAFFIC=profiles/${PFX}_${NAME}_traffic.json
cp $SRC/summary.txt profiles/${PFX}_${NAME}_rocprof_summary.txt
cp $SRC/traffic.json $TRAFFIC
STATS=$(find $SRC/trace -name "*kernel_stats*.csv" 2>/dev/null | head -1)
(head -1 $STATS; grep vstab:: $STATS) > profiles/${PFX}_${NAME}_kernel_stats.csv
python3 - "$PFX" "$NAME" "$KERN" "$TRAFFIC" <<'PY'
import csv, json, sys
pfx, name, kern, traffic = sys.argv[1:5]
rows = list(csv.DictReader(open(f"profiles/{pfx}_{name}_kernel_stats.csv")))
w = [r for r in rows if kern in r["Name"]][0]
t = json.load(open(traffic))
t["rocprof_avg_launch_us"] = round(float(w["AverageNs"]) / 1e3, 2)
json.dump(t, open(traffic, "w"), indent=1)
print(kern, "rocprof avg us:", t["rocprof_avg_launch_us"])
PY
