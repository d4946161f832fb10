#!/bin/bash
# does an initialised process group (RCCL's / gloo's streams in the process) cost the pipeline its fourth hardware queue?  one rank, 4K and 1080p
set -o pipefail
mkdir -p gpurun_out
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'warp', d['roofline']['avg_launch_us'], 'corners wait', s['host_corners_us_per_frame'], d.get('collectives'))"; }
for wl in 4k 1080p; do for rep in 1 2; do for cfg in "plain" "--force-dist --dist-backend nccl" "--force-dist --dist-backend gloo"; do for det in 1 0; do
  a=""; [ "$cfg" != "plain" ] && a="$cfg"
  v=$(VSTAB_DETECT_STREAM=$det timeout -k 10 300 python bench.py --gpus 1 $a --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ab.err | line) || { tail -5 gpurun_out/r04_ab.err; exit 1; }
  echo "$wl [$cfg] detection_stream=$det rep$rep: $v"
done; done; done; done | tee gpurun_out/r04_dist_streams_ab.txt
