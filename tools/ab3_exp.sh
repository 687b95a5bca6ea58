#!/bin/bash
# A/B on one box: library variants; whole frame lone / 4 in flight, 1/8 shard lone / 16 in flight
for lib in "" $1; do
  echo "== lib=${lib:-current}"
  RT_LIB_VARIANT=$lib N_LIST=1 P_LIST=1,4 N_CTX=4 python3 tools/pipeline_cost.py 2>/dev/null | grep shards | cut -c1-330
  GPU_MAX_HW_QUEUES=16 RT_LIB_VARIANT=$lib N_LIST=8 P_LIST=1,16 N_CTX=16 python3 tools/pipeline_cost.py 2>/dev/null | grep shards | cut -c1-330
done
