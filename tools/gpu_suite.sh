#!/bin/bash
# usage (GPU box): bash tools/gpu_suite.sh [tag]   -- the whole -m gpu suite in one process, smoke(), then the driver's bench command twice;
# logs and the two bench lines under gpurun_out/<tag>/
set -o pipefail
TAG=${1:-suite}
O=gpurun_out/$TAG
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; rc=$?
tail -4 $O/gputest.log
[ $rc = 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
for i in 1 2; do
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_default_$i.json 2> $O/bench_default_$i.err || { tail -5 $O/bench_default_$i.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$O/bench_default_$i.json").read().strip().splitlines()[-1])
print("bench $i:", d["value"], d["unit"], "step_ms", d["step_ms"]["median"], d["step_ms"]["min"], d["step_ms"]["max"], "copy_ingest", d.get("copy_ingest", {}).get("value"), "parity", d.get("parity_check"), "roofline frac", d["roofline"]["frac"])
PY
done
