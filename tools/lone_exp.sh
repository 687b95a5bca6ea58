#!/bin/bash
# lone-frame and sharded costs for a parameter setting: tools/lone_exp.sh "name=value,..."
for prm in "" "$1"; do
  echo "== RT_PARAMS=$prm"
  RT_PARAMS=$prm N_LIST=1,8 P_LIST=1,4 N_CTX=4 python3 tools/pipeline_cost.py 2>/dev/null | grep shards
  GPU_MAX_HW_QUEUES=16 RT_PARAMS=$prm N_LIST=8 P_LIST=16 N_CTX=16 python3 tools/pipeline_cost.py 2>/dev/null | grep shards
done
