#!/bin/bash
# compute-side ceiling of the N-GPU split, measured on one GPU: rank 0's shard of an N-way split against the whole frame —
# frame by frame with P slots in flight (round 3's design), and in passes of K frames (rt_trace_shard_batch, round 4)
export GPU_MAX_HW_QUEUES=16
echo "== frame by frame (tools/pipeline_cost.py)"
N_LIST=1 P_LIST=4 N_CTX=4 python3 tools/pipeline_cost.py 2>/dev/null | grep shards | cut -c1-110
for n in 2 4 8; do N_LIST=$n P_LIST=4,16 N_CTX=16 python3 tools/pipeline_cost.py 2>/dev/null | grep shards | cut -c1-120; done
echo "== in passes of K frames (tools/batch_ceiling.py), static frames"
N_LIST=1,2,4,8 K_LIST=1,4,8 P_LIST=4 python3 tools/batch_ceiling.py 2>/dev/null
echo "== in passes of K frames, every frame with its own instances (the reference's animated loop)"
ANIMATE=1 N_LIST=1,8 K_LIST=1,4,8 P_LIST=4 python3 tools/batch_ceiling.py 2>/dev/null
