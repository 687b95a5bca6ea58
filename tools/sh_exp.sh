#!/bin/bash
# shadow-ray entry records: off / rebuilt every frame / kept while light and instances stand still (shadow_entry 0 / 1 / 2)
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "shadow_entry" 2>&1 | tail -3
for prm in "shadow_entry=0" "shadow_entry=2" "shadow_entry=0" "shadow_entry=2"; do
  echo "== $prm"
  RT_PARAMS=$prm N_LIST=1 P_LIST=1,4 N_CTX=4 python3 tools/pipeline_cost.py 2>/dev/null | grep shards | cut -c1-330
  python3 bench.py --param $prm --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   bench ms/step %.4f animated %.4f single %.4f limbs %.4f shadow visits %.2f' % (d['ms_per_step'], d['animated_ms_per_step'], d['ms_per_frame_single'], d['other_mesh']['ms_per_step'], d['roofline']['shadow_kernel']['mean_node_visits_per_ray']))"
done
GPU_MAX_HW_QUEUES=16 RT_PARAMS=shadow_entry=0 N_LIST=8 P_LIST=16 N_CTX=16 python3 tools/pipeline_cost.py 2>/dev/null | grep shards | cut -c1-120
GPU_MAX_HW_QUEUES=16 RT_PARAMS=shadow_entry=2 N_LIST=8 P_LIST=16 N_CTX=16 python3 tools/pipeline_cost.py 2>/dev/null | grep shards | cut -c1-120
