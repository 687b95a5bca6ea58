#!/bin/bash
# compute-side ceiling of the N-GPU split, measured on one GPU: rank 0's shard of an N-way split with P slots in flight against the whole frame
export GPU_MAX_HW_QUEUES=16
N_LIST=1 P_LIST=4 N_CTX=4 python3 tools/pipeline_cost.py 2>/dev/null | grep shards | cut -c1-110
for n in 2 4 8; do N_LIST=$n P_LIST=4,16 N_CTX=16 python3 tools/pipeline_cost.py 2>/dev/null | grep shards | cut -c1-330; done
