#!/bin/bash
# bench.py ms_per_step for a list of "ENV=.. -- bench args" settings: tools/param_sweep.sh <tag> then lines on stdin
TAG=$1; OUT=gpurun_out/psweep_$TAG; mkdir -p $OUT; i=0
while IFS= read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1)); envs="${line%%--*}"; args="${line#*--}"
  env $envs timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras $args > $OUT/r$i.json 2> $OUT/r$i.err
  python3 -c "
import json,sys
try:
    d=json.load(open('$OUT/r$i.json')); k=d['roofline']['frame_kernel_ms']
    print('%-60s ms/step %.4f  closest %.3f shadow %.3f raygen %.3f shade %.3f tail %.3f' % ('''$line'''[:60], d['ms_per_step'], k['trace_closest'], k['trace_shadow'], k['raygen'], k['shade'], k['tail']))
except Exception as e: print('''$line''', 'failed', e)
"
done
