#!/bin/bash
# frame slots in flight on one GPU (HIP's default 4 hardware queues, then 8 queues)
run() { python3 bench.py --frames-in-flight $1 --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('queues ${GPU_MAX_HW_QUEUES:-default} slots $1: ms/step %.4f' % d['ms_per_step'])"; }
unset GPU_MAX_HW_QUEUES
for p in 3 4 5 6 8; do run $p; done
export GPU_MAX_HW_QUEUES=8
for p in 3 4 5 6 8; do run $p; done
