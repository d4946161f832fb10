#!/bin/bash
# usage (GPU box): tools/ab_trace.sh lib1.so lib2.so ...  -- rocprofv3 kernel stats of the 4K pipeline bench with each build of the library
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  OUT=$R/gpurun_out/abt_${lib%.so}
  rm -rf $OUT && mkdir -p $OUT
  VSTAB_LIB_PATH=$R/video-annotator_amd/lib/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/trace.log 2>&1
  python3 - "$OUT" "$lib" <<'PY'
import csv, glob, sys
out, lib = sys.argv[1], sys.argv[2]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_lk_track" in r["Name"] or "k_warp_fused" in r["Name"] or "k_pyr_down" in r["Name"]:
            print(f"{lib:22s} {r['Name'].split('vstab::')[1][:14]:14s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:7.2f} min={float(r['MinNs'])/1e3:7.2f}")
import json
try:
    print(lib, "fps", json.loads(open(out + "/trace.log").read().strip().split("\n")[-1])["value"])
except Exception as e:
    pass
PY
  find $OUT -name "*kernel_trace.csv" -delete
done
