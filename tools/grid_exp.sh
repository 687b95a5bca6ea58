#!/bin/bash
# automatic per-launch caps of the persistent traversal grid (closest_/shadow_blocks_per_cu -1 = default) against no caps (0), 4 frame slots in flight
run() { python3 bench.py $2 --steps 40 --warmup 8 --no-cpu-baseline --no-extras $1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$2 [$1]: ms/step %.4f' % d['ms_per_step'])"; }
OFF="--param closest_blocks_per_cu=0 --param shadow_blocks_per_cu=0"
for wl in "" "--animate" "--mesh limbs" "--workload cfg4" "--workload cfg5"; do
  run "$OFF" "$wl"
  run "" "$wl"
  run "$OFF" "$wl"
  run "" "$wl"
done
