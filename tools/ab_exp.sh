#!/bin/bash
# A/B on one box: library variants x parameter settings; whole frame lone / 4 in flight, 1/8 shard with 16 in flight
for lib in "" old; do for prm in "" "shadow_entry=0" "light_tiles=64" "entry_points=0"; do
  echo "== lib=${lib:-current} params=$prm"
  RT_LIB_VARIANT=$lib RT_PARAMS=$prm N_LIST=1 P_LIST=1,4 N_CTX=4 python3 tools/pipeline_cost.py 2>/dev/null | grep shards | cut -c1-400
  GPU_MAX_HW_QUEUES=16 RT_LIB_VARIANT=$lib RT_PARAMS=$prm N_LIST=8 P_LIST=16 N_CTX=16 python3 tools/pipeline_cost.py 2>/dev/null | grep shards | cut -c1-120
done; done
