#!/bin/bash
# usage (GPU box): bash tools/r02_evidence.sh   -- the round-2 evidence bundle under gpurun_out/r02/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_pipeline_4k.json 2> $O/bench_pipeline_4k.err
python bench.py --workload 1080p --steps 20 --warmup 5 > $O/bench_pipeline_1080p.json 2>> $O/bench.err
python bench.py --steps 20 --warmup 5 --out-format nv12 --no-cpu-baseline > $O/bench_pipeline_4k_nv12_out.json 2>> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-tracking --no-cpu-baseline > $O/bench_undistort_only_4k.json 2>> $O/bench.err
python bench.py --workload 1080p --steps 20 --warmup 5 --no-tracking > $O/bench_undistort_only_1080p.json 2>> $O/bench.err
python bench.py --mode warp --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_warp_only_4k.json 2>> $O/bench.err
python bench.py --workload 4k-p010 --steps 20 --warmup 5 > $O/bench_pipeline_4k_p010_config5.json 2>> $O/bench.err
python tools/quick_p010_time.py > $O/p010_operator_times.txt 2>> $O/bench.err
./tools/probe_rate > $O/valu_issue_rates.txt 2>&1
VSTAB_LIB_PATH=video-annotator_amd/lib/libvstab_dev.so python tools/wg_timeline.py > $O/warp_workgroup_timeline.txt 2>&1
for a in 0 1 2 4 8 3 7 15; do echo "VSTAB_ABLATE=$a"; VSTAB_LIB_PATH=video-annotator_amd/lib/libvstab_dev.so VSTAB_ABLATE=$a python tools/quick_warp_time.py; done > $O/warp_ablations.txt 2>&1
bash tools/prof_warp.sh r02iso > $O/prof_warp.log 2>&1
bash tools/prof_bench.sh r02pipe > $O/prof_bench.log 2>&1
tail -3 $O/prof_bench.log
bash tools/prof_detect.sh > $O/prof_detect.log 2>&1
cp $R/gpurun_out/prof_detect/summary.txt $O/detector_summary.txt
# keep what tools/publish_profiles.sh reads, drop the per-dispatch traces (gpurun merges at most 64 MiB back)
find $R/gpurun_out/prof_r02pipe/trace -name "*kernel_stats.csv" -exec cp {} $O/bench_pipeline_kernel_stats_full.csv \;
find $R/gpurun_out -name "*kernel_trace.csv" -delete
find $R/gpurun_out -name "*counter_collection.csv" -delete
