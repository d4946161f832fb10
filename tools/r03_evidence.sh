#!/bin/bash
# usage (GPU box): tools/r03_evidence.sh  -- everything profiles/history/r03_* is made of (copied there by hand afterwards)
O=gpurun_out/r03ev; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_pipeline_4k.json 2> $O/bench_4k.err
python bench.py --steps 20 --warmup 5 --map-precision opencl --no-cpu-baseline > $O/bench_pipeline_4k_opencl_precision.json 2>> $O/bench_4k.err
python bench.py --steps 20 --warmup 5 --workload 1080p --no-cpu-baseline > $O/bench_pipeline_1080p.json 2>> $O/bench_4k.err
python bench.py --steps 20 --warmup 5 --workload 4k-p010 --no-cpu-baseline > $O/bench_pipeline_4k_p010_config5.json 2>> $O/bench_4k.err
python bench.py --steps 20 --warmup 5 --no-tracking --workload 1080p --no-cpu-baseline > $O/bench_undistort_only_1080p.json 2>> $O/bench_4k.err
python bench.py --steps 20 --warmup 5 --ingest copy --no-cpu-baseline > $O/bench_pipeline_4k_copy_ingest.json 2>> $O/bench_4k.err
tools/prof_warp.sh r03_ieee > /dev/null 2>&1; cp gpurun_out/prof_r03_ieee/summary.txt $O/warp_kernel_isolated_pmc_ieee.txt
QMODE=5 tools/prof_warp.sh r03_ocl > /dev/null 2>&1; cp gpurun_out/prof_r03_ocl/summary.txt $O/warp_kernel_isolated_pmc_opencl.txt
QW=1920 QH=1080 tools/prof_warp.sh r03_1080 > /dev/null 2>&1; cp gpurun_out/prof_r03_1080/summary.txt $O/warp_kernel_isolated_pmc_1080p.txt
bash tools/prof_bench.sh r03bench > $O/prof_bench.log 2>&1; cp gpurun_out/prof_r03bench/summary.txt $O/bench_pipeline_rocprof_summary.txt; cp gpurun_out/prof_r03bench/traffic.json $O/traffic_4k.json
find gpurun_out/prof_r03bench/trace -name "*kernel_stats*.csv" | head -1 | xargs -I{} sh -c '(head -1 {}; grep vstab:: {}) > '$O'/bench_pipeline_kernel_stats.csv'
python tools/wg_timeline.py > $O/warp_phase_timeline_ieee.txt 2>&1
QMODE=5 python tools/wg_timeline.py > $O/warp_phase_timeline_opencl.txt 2>&1
python tools/lk_timeline.py > $O/tracker_feature_timeline.txt 2>&1
tools/prof_ablate.sh > $O/warp_ablations_pmc.txt 2>&1
for s in 1 2; do echo "streams $s"; QSTREAMS=$s python tools/quick_warp_time.py 2>&1 | grep warp; QSTREAMS=$s QMODE=5 python tools/quick_warp_time.py 2>&1 | grep warp; done > $O/warp_two_stream_overlap.txt
rm -rf gpurun_out/prof_r03_ieee gpurun_out/prof_r03_ocl gpurun_out/prof_r03_1080 gpurun_out/prof_r03bench gpurun_out/prof_abl*
ls -la $O
