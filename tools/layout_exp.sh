#!/bin/bash
# node layout A/B (RT_NODE_LAYOUT 0 = builder order, 1 = cache-line treelets), one gpurun call
run() { env "$@" python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['shadow_kernel']; o=d.get('other_mesh',{})
print('$*: ms/step %.4f animated %.4f single %.4f limbs %.4f | closest iso %.3f lone %.3f | shadow live %.3f' % (d['ms_per_step'], d.get('animated_ms_per_step',0), d.get('ms_per_frame_single',0), o.get('ms_per_step',0), r['isolated']['avg_launch_ms'], r['isolated_lone_slot']['avg_launch_ms'], s['avg_launch_ms']))"; }
run RT_NODE_LAYOUT=0
run RT_NODE_LAYOUT=1
run RT_NODE_LAYOUT=0
run RT_NODE_LAYOUT=1
for l in 0 1; do RT_NODE_LAYOUT=$l N_LIST=1 P_LIST=1 N_CTX=1 python3 tools/pipeline_cost.py 2>/dev/null | grep shards | cut -c1-330; done
