#!/bin/bash
# Runs bench.py (no extras, no CPU leg) once per experimental library build: tools/exp_sweep.sh <tag> "<variants>" [bench args]
TAG=$1; VARS=$2; shift 2
OUT=gpurun_out/sweep_$TAG; mkdir -p $OUT
for v in $VARS; do
  for P in 1 4; do
    RT_LIB_VARIANT=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --frames-in-flight $P "$@" > $OUT/${v}_p$P.json 2> $OUT/${v}_p$P.err || echo "$v p$P failed" >> $OUT/progress.log
    echo "$v p$P done" >> $OUT/progress.log
  done
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; k=r["frame_kernel_ms"]
        print("%-22s ms/step %.4f  closest live %.3f iso %.3f  shadow %.3f raygen %.3f shade %.3f tail %.3f"%(os.path.basename(f)[:-5], d["ms_per_step"], k["trace_closest"], r["isolated"]["avg_launch_ms"], k["trace_shadow"], k["raygen"], k["shade"], k["tail"]))
    except Exception as e: print(f, "unreadable", e)
PY
