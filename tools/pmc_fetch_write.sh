cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for a in 0 256; do
  for c in FETCH_SIZE WRITE_SIZE; do
    VSTAB_ABLATE=$a rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmcfw_${a}_$c -- python3 $R/tools/quick_warp_time.py > /dev/null 2>&1
    python3 - <<PY
import csv,glob
v=[float(r["Counter_Value"]) for f in glob.glob("$R/gpurun_out/pmcfw_${a}_$c/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "k_warp_fused" in r["Kernel_Name"]]
print("ablate $a $c mean KB", sum(v)/len(v), len(v))
PY
  done
done
