#!/bin/bash
# 1/8 shard with 16 slots in flight (and lone) for parameter settings: tools/shard_param_exp.sh "a=1 b=2,c=3 ..."
export GPU_MAX_HW_QUEUES=${QUEUES:-16}
for prm in "" $1; do
  echo "== RT_PARAMS=$prm"
  RT_PARAMS=$prm N_LIST=8 P_LIST=1,16 N_CTX=16 python3 tools/pipeline_cost.py 2>/dev/null | grep shards | cut -c1-330
done
