#!/bin/bash
mkdir -p gpurun_out
for cfg in "--workload 4k" "--workload 1080p" "--workload 4k --ingest copy" "--workload 4k-p010"; do
  echo "== $cfg"
  VSTAB_HOST_TIMING=1 timeout -k 10 200 python bench.py $cfg --steps 40 --warmup 5 --no-cpu-baseline --skip-copy-pass --skip-ieee-pass 2>gpurun_out/r04_ht.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['parity_check'])"
  grep -i -A30 "host timing\|HOST_TIMING\|us per\|per frame" gpurun_out/r04_ht.err | head -45
done 2>&1 | tee gpurun_out/r04_host_timing.txt
