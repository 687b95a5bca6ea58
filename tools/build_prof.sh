#!/bin/bash
# per-kernel times of the device BLAS builders (cfg3 scene): tools/build_prof.sh [algo]
export TMPDIR=/tmp
OUT=gpurun_out/prof_build; rm -rf $OUT; mkdir -p $OUT
RT_BUILD_TIMING=1 RT_GPU_BVH_ALGO=${1:-3} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/build_profile.py > $OUT/run.log 2> $OUT/run.err
grep bvh_gpu $OUT/run.err | head -8
f=$(find $OUT -name "*kernel_stats.csv" | head -1); echo $f; head -24 "$f" | cut -c1-160
