#!/bin/bash
# round 4 evidence: bench lines of every configuration, sustained rate, rocprofv3 kernel trace + PMC passes of the default command
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r04
O=gpurun_out/r04
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }; python -c "import json,sys; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['parity_check'], d['roofline']['avg_launch_us'], d['roofline'].get('alone'))"; }
run bench_pipeline_4k --steps 20 --warmup 5
run bench_pipeline_4k_second_run --steps 20 --warmup 5
run bench_pipeline_4k_ieee_map --steps 20 --warmup 5 --map-precision ieee --no-cpu-baseline --skip-copy-pass
run bench_pipeline_4k_pull_batch --steps 20 --warmup 5 --pull batch --no-cpu-baseline --skip-copy-pass
run bench_pipeline_4k_nv12_out --steps 20 --warmup 5 --out-format nv12 --no-cpu-baseline --skip-copy-pass
run bench_pipeline_1080p --workload 1080p --steps 20 --warmup 5
run bench_undistort_only_1080p --workload 1080p --no-tracking --steps 20 --warmup 5 --no-cpu-baseline
run bench_pipeline_4k_p010_config5 --workload 4k-p010 --steps 20 --warmup 5
run bench_pipeline_4k_p010_config5_p010_out --workload 4k-p010 --out-format p010 --steps 20 --warmup 5 --no-cpu-baseline
run bench_warp_only_4k --mode warp --steps 20 --warmup 5 --no-cpu-baseline
run sustained_rate_192k_frames --steps 3000 --warmup 5 --no-cpu-baseline --skip-copy-pass
ROUND=r04 MAP_PRECISION=opencl bash tools/prof_bench.sh r04 > $O/prof_bench.log 2>&1 || { tail -20 $O/prof_bench.log; exit 1; }
cp gpurun_out/prof_r04/summary.txt $O/bench_pipeline_rocprof_summary.txt
cp gpurun_out/prof_r04/traffic.json $O/traffic_4k.json
f=$(find gpurun_out/prof_r04/trace -name "*kernel_stats.csv" | head -1); cp $f $O/bench_pipeline_kernel_stats.csv
rm -rf gpurun_out/prof_r04/trace gpurun_out/prof_r04/pmc_* gpurun_out/prof_r04/cal_*   # raw counter dumps: tens of MB
grep "^stats\|^traffic" $O/bench_pipeline_rocprof_summary.txt
