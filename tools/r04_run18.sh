#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for wl in 1080p 4k; do
VSTAB_HOST_TIMING=1 timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass > gpurun_out/r04_ht_$wl.json 2>gpurun_out/r04_ht_$wl.err || { tail -5 gpurun_out/r04_ht_$wl.err; exit 1; }
python -c "import sys,json; d=json.loads(open('gpurun_out/r04_ht_$wl.json').read().strip().splitlines()[-1]); print('$wl', d['value'], d['stages_timed_region'])"
grep -i "host timing\|pull_cb\|ingest\|pyramid\|spec_detect\|lk_\|warp\|pull_frame_total" gpurun_out/r04_ht_$wl.err | head -20
done
