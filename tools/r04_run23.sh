#!/bin/bash
# run-to-run spread of the default bench line (4K): 12 runs on one box
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2 3 4 5 6 7 8 9 10 11 12; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass > gpurun_out/r04_sp_$rep.json 2>gpurun_out/r04_sp_$rep.err || { tail -5 gpurun_out/r04_sp_$rep.err; exit 1; }
  python -c "import json; d=json.loads(open('gpurun_out/r04_sp_$rep.json').read().strip().splitlines()[-1]); s=d['stages_timed_region']; print('rep $rep', d['value'], 'preroll', d['preroll'], 'warp', d['roofline']['avg_launch_us'], 'wait', s['host_track_wait_us_per_frame'], 'corners', s['host_corners_us_per_frame'], 'est', s['host_estimate_us_per_frame'], 'keys', s['key_frames'])"
done
