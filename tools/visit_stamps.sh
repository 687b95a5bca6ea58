#!/bin/bash
# what a wave's fast visit is made of (RT_EXP_VISIT_STAMPS builds, instrumented kernels): tools/visit_stamps.sh
for k in 1 2 3; do echo "== stamps $k (1: node fetch issue -> arrival, 2: arrival -> end of visit, 3: whole visit)"; RT_LIB_VARIANT=vs$k python3 tools/trace_bench.py --variants 0 --blocks 5 --frames 2 --blas-builder 1 2>/dev/null | grep diag | cut -c1-400; done
echo "== default counting build"; python3 tools/trace_bench.py --variants 0 --blocks 5 --frames 2 --blas-builder 1 2>/dev/null | grep -E "diag|frame_ms" | cut -c1-400
