#!/bin/bash
# the speculative corner detection's latency and the host's wait for corners at key frames (VSTAB_DEBUG_SPEC=1), consecutive runs
mkdir -p gpurun_out
for wl in 4k 4k 4k 1080p 4k-p010; do
echo "== $wl"
VSTAB_DEBUG_SPEC=1 timeout -k 10 200 python bench.py --workload $wl --steps 40 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_spec.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_timed_region']; print(d['value'], d['parity_check'], 'corners wait', s['host_corners_us_per_frame'], 'track wait', s['host_track_wait_us_per_frame'])"
grep "async selection" gpurun_out/r04_spec.err | awk '{print $4}' | sort -n | awk '{a[NR]=$1} END {print "detection latency us (helper thread): n", NR, "min", a[1], "median", a[int(NR/2)], "p90", a[int(NR*0.9)], "max", a[NR]}'
echo "selections done by the caller: $(grep -c 'selection done by the caller' gpurun_out/r04_spec.err)"
grep "key frames pre-launched" gpurun_out/r04_spec.err | sed 's/.*key frames pre-launched/key frames pre-launched/'
done 2>&1 | tee gpurun_out/r04_keyframe_debug3.txt
