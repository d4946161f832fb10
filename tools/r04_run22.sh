#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
O=gpurun_out/r04
for rep in 1 2 3; do
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --out-format nv12 --no-cpu-baseline --skip-copy-pass > $O/nv12_$rep.json 2> $O/nv12_$rep.err || { tail -5 $O/nv12_$rep.err; exit 1; }
python -c "import json; d=json.loads(open('$O/nv12_$rep.json').read().strip().splitlines()[-1]); print('nv12 out', d['value'], d['parity_check'], d['roofline']['avg_launch_us'], d['roofline']['alone']['avg_launch_us'])"
done
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/gputest_final.log 2>&1; rc=$?
tail -4 $O/gputest_final.log
exit $rc
