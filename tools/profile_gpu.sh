#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats and PMC traffic passes for bench.py.
# Usage: tools/profile_gpu.sh <tag> [bench args]      -> gpurun_out/prof_<tag>/...
set -o pipefail
TAG=${1:-r01}
shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline $@"   # the default command (60 steps, 4 frames in flight), minus the CPU leg
# 1) per-kernel time
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || { echo "kernel-trace failed"; tail -5 $OUT/trace.log; exit 1; }
# 2) PMC passes (own runs, no tracing domains besides kernel-trace): FETCH_SIZE and WRITE_SIZE separately
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $OUT/pmc_write.log; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- $BENCH > $OUT/pmc_l2.log 2>&1 || echo "pmc l2 failed (non-fatal)"
find $OUT -name "*.csv" | head -40
python3 tools/summarize_profile.py $OUT --json $OUT/summary.json > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
