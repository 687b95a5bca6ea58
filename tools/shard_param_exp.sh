#!/bin/bash
# per-launch grid caps on shards: rank 0's 1/2 and 1/4 shard (4 and 16 slots) with the automatic caps and without
for n in 2 4; do for prm in "" "closest_blocks_per_cu=0,shadow_blocks_per_cu=0" "" "closest_blocks_per_cu=0,shadow_blocks_per_cu=0"; do
  echo -n "N=$n [$prm] "; GPU_MAX_HW_QUEUES=16 RT_PARAMS=$prm N_LIST=$n P_LIST=4,16 N_CTX=16 python3 tools/pipeline_cost.py 2>/dev/null | grep shards | cut -c1-70 | tr '\n' ' '; echo
done; done
