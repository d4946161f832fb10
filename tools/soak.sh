#!/bin/bash
# long runs of every pipeline configuration (parity check of an emitted frame at the end of each), then the whole -m gpu suite once more
set -o pipefail
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['parity_check'], d['steps'] * d['config']['frames_per_step'], 'frames')"; }
for cfg in "--workload 1080p --steps 3000" "--workload 4k --steps 1500 --ingest copy" "--workload 4k-p010 --steps 1000" "--workload 1080p --steps 1500 --pull batch" "--workload 4k --steps 1000 --out-format nv12"; do
  v=$(timeout -k 10 300 python bench.py $cfg --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "$cfg: $v"
done | tee gpurun_out/r04_soak.txt
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputest_s3.log 2>&1; rc=$?
tail -3 gpurun_out/r04_gputest_s3.log
exit $rc
