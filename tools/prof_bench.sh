#!/bin/bash
# usage (on the GPU box): [EXTRA="--out-format nv12-planar" WARP_KERNEL=k_warp_planar] bash tools/prof_bench.sh <tag>
# kernel-trace stats + separate PMC passes (FETCH_SIZE / WRITE_SIZE each alone) of the default bench
# command (plus EXTRA arguments), plus the PMC calibration kernels.  Output: gpurun_out/prof_<tag>/
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-bench}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="$R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --skip-copy-pass $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- python3 $R/tools/calib_traffic.py > $OUT/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -- python3 $R/tools/calib_traffic.py > $OUT/cal_write.log 2>&1
python3 $R/tools/summarize_bench_prof.py $OUT
# the per-dispatch tables are tens of MB each (gpurun copies back at most 64 MiB): keep the statistics, the summary and the logs
find $OUT/trace -name "*kernel_trace.csv" -delete
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/cal_fetch $OUT/cal_write
du -sh $OUT
