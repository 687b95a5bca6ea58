#!/bin/bash
# BLAS builder comparison on the headline workload, one gpurun call: device LBVH, device binned SAH, host binned SAH (threaded)
run() { env "$@" RT_BUILD_TIMING=1 python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>gpurun_out/builder_exp.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['shadow_kernel']; k=r['frame_kernel_ms']; o=d.get('other_mesh',{})
print('$*: ms/step %.4f animated %.4f single %.4f limbs %.4f | closest nodes %.2f tris %.2f iso %.3f lone %.3f | shadow nodes %.2f tris %.2f live %.3f' % (d['ms_per_step'], d.get('animated_ms_per_step',0), d.get('ms_per_frame_single',0), o.get('ms_per_step',0), r['mean_node_visits_per_ray'], r['mean_tri_tests_per_ray'], r['isolated']['avg_launch_ms'], r['isolated_lone_slot']['avg_launch_ms'], s['mean_node_visits_per_ray'], s['mean_tri_tests_per_ray'], s['avg_launch_ms']))"; grep "bvh_build\] n 34\|bvh_gpu" gpurun_out/builder_exp.err | sort | uniq -c | sort -rn | head -4; }
run RT_GPU_BVH_ALGO=1
run RT_GPU_BVH_ALGO=3
run RT_BLAS_BUILDER=0 RT_BVH_MAX_LEAF=1
run RT_GPU_BVH_ALGO=1
run RT_GPU_BVH_ALGO=3
