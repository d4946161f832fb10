#!/bin/bash
# the bench lines that carry CPU baselines (run again after the OpenMP-pool fix), then the whole -m gpu suite
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r04
O=gpurun_out/r04
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }; python -c "import json,sys; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['parity_check'], d['roofline']['avg_launch_us'], {k:(v['value'],v['cores']) for k,v in d.items() if k.startswith('cpu_baseline')})"; }
run bench_pipeline_4k --steps 20 --warmup 5
run bench_pipeline_4k_second_run --steps 20 --warmup 5
run bench_pipeline_1080p --workload 1080p --steps 20 --warmup 5
run bench_pipeline_4k_p010_config5 --workload 4k-p010 --steps 20 --warmup 5
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/gputest_final.log 2>&1; rc=$?
tail -4 $O/gputest_final.log
exit $rc
