#!/bin/bash
# occupancy experiment: library variants built with other waves-per-SIMD / LDS budgets, lone frame and 4 frames in flight
run() { RT_LIB_VARIANT=$1 RT_TRACE_BLOCKS_PER_CU=$2 python3 bench.py --no-cpu-baseline --no-extras --frames-in-flight $3 --steps 40 --warmup 8 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['frame_kernel_ms']; print('$1 blocks/CU $2 slots $3: ms/step %.4f closest %.3f shadow %.3f' % (d['ms_per_step'], k['trace_closest'], k['trace_shadow']))"; }
for v in $1; do run $v 6 1; run $v 5 1; run $v 4 4; run $v 3 4; run $v 2 4; done
