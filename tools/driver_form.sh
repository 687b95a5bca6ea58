#!/bin/bash
# the driver's invocation (--steps 20 --warmup 5) against longer runs, alternating, one gpurun call
for a in "--steps 20 --warmup 5" "--steps 60 --warmup 6" "--steps 20 --warmup 5" "--steps 60 --warmup 6" "--steps 20 --warmup 12"; do python3 bench.py --gpus 1 $a --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$a: ms/step %.4f value %.0f' % (d['ms_per_step'], d['value']))"; done
