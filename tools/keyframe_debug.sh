#!/bin/bash
mkdir -p gpurun_out
for wl in 4k 4k-p010; do for pf in 8 12 16; do
echo "== $wl prefetch $pf"
VSTAB_PREFETCH=$pf VSTAB_DEBUG_SPEC=1 timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_spec.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'corners wait', s['host_corners_us_per_frame'], 'track wait', s['host_track_wait_us_per_frame'])"
grep "async selection" gpurun_out/r04_spec.err | awk '{s+=$4; n++} END {print "detection wait mean us:", s/n, "over", n}'
grep "key frames pre-launched" gpurun_out/r04_spec.err | sed 's/.*key frames pre-launched/key frames pre-launched/'
done; done 2>&1 | tee gpurun_out/r04_keyframe_debug.txt
