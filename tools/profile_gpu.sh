#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats and PMC traffic passes for bench.py.
# Usage: tools/profile_gpu.sh <tag> [bench args]      -> gpurun_out/prof_<tag>/...  (summary.txt, latest_profile.json, bench_under_profiler.json)
# The program follows `--` directly; counters are collected in their own passes with --kernel-trace only.
set -o pipefail
TAG=${1:-r02}
shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extras $@"
# 1) per-kernel time: the bench line printed under the profiler is kept next to the trace it belongs to
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_under_profiler.json 2> $OUT/trace.log || { echo "kernel-trace failed"; tail -5 $OUT/trace.log; exit 1; }
# 2) PMC passes (own runs, no tracing domains besides kernel-trace): FETCH_SIZE and WRITE_SIZE separately
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS --steps 8 > $OUT/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS --steps 8 > $OUT/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $OUT/pmc_write.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 bench.py $ARGS --steps 8 > $OUT/pmc_l2.log 2>&1 || echo "pmc l2 failed (non-fatal)"
export RT_PROFILE_TAG=$(python3 -c "
import json,sys,bench
a=' $ARGS '.split()
def opt(n,d):
    return a[a.index(n)+1] if n in a else d
print(json.dumps({'workload':opt('--workload','cfg3'),'mesh':opt('--mesh','standin'),'variant':int(opt('--variant',0)),'n_gpus':1,'frames_in_flight':int(opt('--frames-in-flight',4)),'kernels_sha16':bench.kernels_sha16()}))")
export RT_PROFILE_SOURCE="profiles/${TAG}_rocprof_summary.txt (rocprofv3 --kernel-trace --stats -- python3 bench.py $ARGS)"
WL=$(python3 -c "
a=' $ARGS '.split()
print(a[a.index('--workload')+1] if '--workload' in a else 'cfg3')")
LATEST=latest_profile.json; [ "$WL" != "cfg3" ] && LATEST=latest_profile_$WL.json
python3 tools/summarize_profile.py $OUT --json $OUT/summary.json --latest $OUT/$LATEST > $OUT/summary.txt 2>&1
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f 0.3 >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
