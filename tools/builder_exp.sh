#!/bin/bash
# BLAS builder comparison on the headline workload: device LBVH (default), device PLOC, host binned SAH with one triangle per leaf
run() { env "$@" python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['shadow_kernel']; print('$*: ms/step %.4f  closest nodes %.2f tris %.2f  shadow nodes %.2f tris %.2f' % (d['ms_per_step'], r['mean_node_visits_per_ray'], r['mean_tri_tests_per_ray'], s['mean_node_visits_per_ray'], s['mean_tri_tests_per_ray']))"; }
run RT_X=0
run RT_GPU_BVH_ALGO=2
run RT_BLAS_BUILDER=0 RT_BVH_MAX_LEAF=1
run RT_LBVH_ROTATE=4
