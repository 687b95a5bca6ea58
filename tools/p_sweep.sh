#!/bin/bash
# frames in flight x hardware queues on the headline workload: tools/p_sweep.sh <tag> [bench args]
TAG=$1; shift
OUT=gpurun_out/psweep_$TAG; mkdir -p $OUT
for Q in 4 8; do for P in 2 3 4 6 8; do
  GPU_MAX_HW_QUEUES=$Q timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --frames-in-flight $P --steps 48 --warmup 8 "$@" > $OUT/q${Q}_p$P.json 2> $OUT/q${Q}_p$P.err
  python3 -c "
import json; d=json.load(open('$OUT/q${Q}_p$P.json')); print('queues $Q slots $P: ms/step %.4f  %.0f Mrays/s' % (d['ms_per_step'], d['value']))"
done; done
