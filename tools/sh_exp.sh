#!/bin/bash
# kept shadow-ray entry records (shadow_entry 2): tiles per side of the cube around the light
for prm in "light_tiles=128" "light_tiles=256" "light_tiles=384" "light_tiles=64" "light_tiles=128"; do
  echo "== $prm"
  RT_PARAMS=$prm N_LIST=1 P_LIST=1,4 N_CTX=4 python3 tools/pipeline_cost.py 2>/dev/null | grep shards | cut -c1-330
  RT_PARAMS=$prm MESHES=standin python3 tools/visit_counts.py 2>/dev/null | grep workload | cut -c150-400
done
